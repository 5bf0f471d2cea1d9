// ht_api.hip -- the C-ABI layer (include/ht_mi355x.h): context, device memory, launch sequencing.
// Host code stays C++ and only sequences hand-written HIP kernels; there is no CPU fallback: every entry point fails with
// HT_ERR_HIP when no gfx950 device is usable.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include "ht_device.hpp"
#include "ht_host.hpp"
#include "ht_launch.hpp"
#include "ht_model_build.hpp"

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return HT_ERR_HIP; } } while (0)

#define CHECK_READY(ctx) if (!(ctx)) return HT_ERR_ARG; if (!(ctx)->ready) { (ctx)->err = "context not initialised (ht_create failed)"; return HT_ERR_STATE; } ht_device_guard dev_guard_((ctx)->device)
#define CHECK_MODEL(ctx) do { if ((ctx)->cnn_only) { (ctx)->err = "this context was created without a hand model (CNN only)"; return HT_ERR_STATE; } } while (0)
#define CHECK_BATCH(ctx, B) do { if ((B) < 1 || (B) > (ctx)->B) { (ctx)->err = "batch exceeds the capacity given to ht_create"; return HT_ERR_ARG; } } while (0)

// ------------------------------------------------------------------------------------------------- context
static void default_params(ht_params &p)     // handtrack.h:523-547, physics.h:45-47, physmodel.h:234, handtrack.h:369,450
{
	p.full_reset_on_error = 0.6f; p.angles_only = 0; p.always_take_cnn = 0; p.drangey = 0.7f; p.boundary_planes = 1; p.microforce = 1.0f;
	p.cloudforce_max_point = 15.0f; p.cloudforce_max_sum = 3000.0f; p.mainthreadpasses = 1; p.subsample_fraction = 4; p.min_point_num = 400; p.subsample_voxel = 0; p.subsample_size = 0.0f;
	p.accum_error_threshold = 0.0f; p.min_cray_prob = 0.0f; p.steps = 5; p.steps_keypoints = 3; p.steps_keyangles = 2; p.steps_palmangle = 2; p.steps_cloudstart = 1; p.steps_unibody = 3;
	p.physics_iterations = 16; p.physics_iterations_post = 4; p.physics_use_collision = 1; p.physics_weak_force = 0.4f; p.bone_sum_error_scale = 4.0f; p.unibody_force = 0.1f;
}

template <class T> static int dev_alloc(ht_ctx *ctx, T **p, size_t n)
{
	void *q = nullptr;
	HIPCHK(ctx, hipMalloc(&q, n * sizeof(T)));
	ctx->allocs.push_back(q);
	*p = (T *)q;
	return HT_OK;
}
template <class T> static int dev_upload(ht_ctx *ctx, T **p, const std::vector<T> &h)
{
	int r = dev_alloc(ctx, p, h.size() ? h.size() : 1);
	if (r) return r;
	if (h.size()) HIPCHK(ctx, hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
	return HT_OK;
}

// The contact kernel's image of the collision vertices: every body padded to whole rows of 16 with copies of its vertex 0 (index 0), built from
// the host copy (again after ht_scale).
static int upload_padded_verts(ht_ctx *ctx)
{
	ht_model_dev &m = ctx->model;
	std::vector<float4> pv;
	for (int b = 0; b < m.nb; b++)
	{
		m.cvert_off[b] = (int)pv.size();
		const int n = m.vert_off[b + 1] - m.vert_off[b], np = (n + 15) & ~15;
		for (int k = 0; k < np; k++) { const int src = k < n ? k : 0; float4 q = ctx->h_verts[(size_t)m.vert_off[b] + src]; memcpy(&q.w, &src, 4); pv.push_back(q); }
	}
	m.cvert_off[m.nb] = (int)pv.size();
	if (!ctx->d_cverts_rw) { int r = dev_alloc(ctx, &ctx->d_cverts_rw, pv.size() ? pv.size() : 1); if (r) return r; }
	if (pv.size()) HIPCHK(ctx, hipMemcpy(ctx->d_cverts_rw, pv.data(), pv.size() * sizeof(float4), hipMemcpyHostToDevice));
	m.cverts = ctx->d_cverts_rw;
	return HT_OK;
}

static int load_model(ht_ctx *ctx, const char *path)
{
	// `path` is either the reference's model JSON (built here, a33) or a model baked earlier with ht_model_bake
	fx_map fx;
	char magic[8] = { 0 };
	{ FILE *fp = fopen(path, "rb"); if (!fp) { ctx->err = std::string("cannot open model ") + path; return HT_ERR_IO; } size_t n = fread(magic, 1, 8, fp); (void)n; fclose(fp); }
	if (!memcmp(magic, "HTFX0001", 8)) { if (!fx_load(path, fx)) { ctx->err = std::string("cannot read baked model ") + path; return HT_ERR_IO; } }
	else if (!ht_build_model(path, HT_BUILD_HAND_TWEAKS, fx, ctx->err)) return HT_ERR_IO;
	auto need = [&](const char *n) -> const fx_arr * { auto it = fx.find(n); if (it == fx.end()) { ctx->err = std::string("model entry missing: ") + n; return nullptr; } return &it->second; };
	const fx_arr *a;
	if (!(a = need("nb"))) return HT_ERR_IO; int nb = a->i()[0];
	if (!(a = need("nj"))) return HT_ERR_IO; int nj = a->i()[0];
	// body slot HT_MAXNB-1 is the solver's idle body; 26 joints: the most whose angular rows a solve can always keep (13 CNN-driven + 6 range rows + slowfit's 3 relative
	// rows per joint = 247 of 252: csrc/ht_solver.hip), so that no model this accepts can run into that capacity
	if (nb < 1 || nb >= HT_MAXNB || nj > 26) { ctx->err = "model too large (at most 31 bodies and 26 joints)"; return HT_ERR_IO; }
	ht_model_dev &m = ctx->model;
	memset(&m, 0, sizeof m);
	m.nb = nb; m.nj = nj;
	const fx_arr *bf = need("body_f"), *bc = need("body_collide"), *ig = need("ignore"), *ji = need("joint_i"), *jf = need("joint_f"), *ph = need("physics"), *ubv = need("unibody/verts"), *ubf = need("unibody/f");
	if (!bf || !bc || !ig || !ji || !jf || !ph || !ubv || !ubf) return HT_ERR_IO;
	const float *P = ph->f();
	ht_physics_dev &phys = ctx->phys;
	phys.deltaT = P[0]; phys.restitution = P[1]; phys.gravity_len = sqrtf((P[2] * P[2] + P[3] * P[3]) + P[4] * P[4]); phys.coloumb = P[5]; phys.biasfactorjoint = P[6]; phys.biasfactorpositive = P[7];
	phys.falltime_to_ballistic = P[9]; phys.driftmax = P[10];
	phys.cos40d = cos((double)(40.0f * 3.14f / 180.0f));      // handtrack.h:437 (cos resolves to the double overload there)
	phys.cos40 = (float)phys.cos40d;
	phys.jiggle_sin = sinf(3.14f / 180.0f * (4.0f) / 2.0f);    // gjk.h:626-628
	const float physics_damping = P[11];
	std::vector<float4> verts, planes; std::vector<float> bodyc((size_t)nb * HT_BC, 0.0f), jointc((size_t)nj * HT_JC, 0.0f);
	ctx->h_bodyc.clear();
	for (int b = 0; b < nb; b++)
	{
		char nm[64];
		snprintf(nm, sizeof nm, "b%d/verts", b); const fx_arr *v = need(nm); if (!v) return HT_ERR_IO;
		snprintf(nm, sizeof nm, "b%d/planes", b); const fx_arr *pl = need(nm); if (!pl) return HT_ERR_IO;
		m.vert_off[b] = (int)verts.size(); m.plane_off[b] = (int)planes.size();
		float diam2 = 0.0f;
		for (uint32_t i = 0; i < v->dims[0]; i++) { float4 q = make_float4(v->f()[3 * i], v->f()[3 * i + 1], v->f()[3 * i + 2], 0.0f); memcpy(&q.w, &i, 4); verts.push_back(q); }      // w = index within the body (bit pattern), used by the support scans
		for (uint32_t i = 0; i < v->dims[0]; i++) for (uint32_t j = i + 1; j < v->dims[0]; j++)
		{
			float dx = v->f()[3 * i] - v->f()[3 * j], dy = v->f()[3 * i + 1] - v->f()[3 * j + 1], dz = v->f()[3 * i + 2] - v->f()[3 * j + 2];
			float d2 = dx * dx + dy * dy + dz * dz; if (d2 > diam2) diam2 = d2;
		}
		for (uint32_t i = 0; i < pl->dims[0]; i++) planes.push_back(make_float4(pl->f()[4 * i], pl->f()[4 * i + 1], pl->f()[4 * i + 2], pl->f()[4 * i + 3]));
		const float *r = bf->f() + 26 * b;     // mass massinv radius radius_inner damping friction gravscale com3 pos_start3 quat_start4 tensorinv9
		float *c = &bodyc[(size_t)b * HT_BC];
		c[HT_BC_MASS] = r[0]; c[HT_BC_MASSINV] = r[1]; c[HT_BC_RADIUS] = r[2]; c[HT_BC_RINNER] = r[3];
		c[HT_BC_DAMPLEFT] = powf((1.0f - (r[4] < physics_damping ? physics_damping : r[4])), phys.deltaT);      // physics.h:506
		c[HT_BC_FRICTION] = r[5]; c[HT_BC_DIAM] = sqrtf(diam2);
		for (int i = 0; i < 3; i++) c[HT_BC_COM + i] = r[7 + i];
		for (int i = 0; i < 3; i++) c[HT_BC_POS0 + i] = r[10 + i];
		for (int i = 0; i < 4; i++) c[HT_BC_Q0 + i] = r[13 + i];
		for (int i = 0; i < 9; i++) c[HT_BC_TINV + i] = r[17 + i];
		m.collide[b] = bc->i()[b];
		unsigned mask = 0; for (int j = 0; j < nb; j++) if (ig->i()[b * nb + j]) mask |= 1u << j;
		m.ignore[b] = mask;
	}
	m.vert_off[nb] = (int)verts.size(); m.plane_off[nb] = (int)planes.size();
	// HandModelEnhancements' one-time rewrite (handtrack.h:408-416): a model whose bone 2 (the thumb base) ignores fewer than 10 bodies gets that
	// bone out of every collision pair.  The reference does it on the first call, which precedes every solve with collisions (:684-685, :773-779,
	// :798), so doing it at load gives the same pairs.  The list length counts duplicates there; a baked file without "ignore_count" falls back
	// to the number of distinct bodies.
	if (nb > 2)
	{
		auto itc = fx.find("ignore_count");
		int n2 = 0;
		if (itc != fx.end()) n2 = itc->second.i()[2]; else for (int j = 0; j < nb; j++) n2 += (m.ignore[2] >> j) & 1u;
		if (n2 < 10)
		{
			m.ignore[2] = 0;
			for (int j = 0; j < nb; j++) if (j != 2) { m.ignore[2] |= 1u << j; m.ignore[j] |= 1u << 2; }
		}
	}
	for (int j = 0; j < nj; j++)
	{
		float *c = &jointc[(size_t)j * HT_JC]; const float *r = jf->f() + 16 * j;
		c[HT_JC_RB0] = (float)ji->i()[2 * j]; c[HT_JC_RB1] = (float)ji->i()[2 * j + 1];
		for (int i = 0; i < 3; i++) { c[HT_JC_P0 + i] = r[i]; c[HT_JC_P1 + i] = r[3 + i]; c[HT_JC_RMIN + i] = r[6 + i]; c[HT_JC_RMAX + i] = r[9 + i]; }
		for (int i = 0; i < 4; i++) c[HT_JC_FRAME + i] = r[12 + i];
	}
	{   // unibody cube: f = mass massinv radius damping friction gravscale com3 tensorinv9
		const float *u = ubf->f();
		m.ub_massinv = u[1];
		m.ub_dampleft = powf((1.0f - (u[3] < physics_damping ? physics_damping : u[3])), phys.deltaT);
		for (int i = 0; i < 3; i++) m.ub_com[i] = u[6 + i];
		for (int i = 0; i < 9; i++) m.ub_tinv[i] = u[9 + i];
	}
	ctx->h_bodyc = bodyc; ctx->h_jointc = jointc;
	float4 *dv, *dp; float *dbc, *djc;
	int r;
	if ((r = dev_upload(ctx, &dv, verts)) || (r = dev_upload(ctx, &dp, planes)) || (r = dev_upload(ctx, &dbc, bodyc)) || (r = dev_upload(ctx, &djc, jointc))) return r;
	m.verts = dv; m.planes = dp; m.bodyc = dbc; m.jointc = djc;
	ctx->h_verts = verts; ctx->h_planes = planes; ctx->d_verts_rw = dv; ctx->d_planes_rw = dp; ctx->d_bodyc_rw = dbc; ctx->d_jointc_rw = djc;
	return upload_padded_verts(ctx);
}

static void sync_params(ht_ctx *ctx)
{
	ctx->phys.iterations = ctx->par.physics_iterations; ctx->phys.iterations_post = ctx->par.physics_iterations_post; ctx->phys.use_collision = ctx->par.physics_use_collision;
	ctx->phys.weak_force = ctx->par.physics_weak_force; ctx->phys.bone_sum_error_scale = ctx->par.bone_sum_error_scale; ctx->phys.unibody_force = ctx->par.unibody_force;
}

extern "C" int ht_create(const char *model_path, int max_batch, int device, ht_ctx **out)
{
	if (!out || max_batch < 1) return HT_ERR_ARG;
	ht_ctx *ctx = new ht_ctx();
	*out = ctx;      // returned even on failure so that ht_last_error can be read; caller destroys it
	ctx->B = max_batch; ctx->device = device;
	default_params(ctx->par);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { ctx->err = "no HIP device: the MI355X hot path has no CPU fallback"; return HT_ERR_HIP; }
	if (device < 0 || device >= ndev) { ctx->err = "no such HIP device"; return HT_ERR_ARG; }
	ht_device_guard dev_guard_(device);      // the caller's current device is restored on return
	hipDeviceProp_t prop;
	HIPCHK(ctx, hipGetDeviceProperties(&prop, device));
	if (!strstr(prop.gcnArchName, "gfx950")) { ctx->err = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only"; return HT_ERR_HIP; }
	HIPCHK(ctx, hipStreamCreate(&ctx->stream));
	for (int i = 0; i < 2; i++) { HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking)); HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming)); }
	HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
	HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_lap, hipEventDisableTiming));
	// model_path == NULL: a context that only evaluates / trains the CNN (what a stand-alone CNN object of the reference is, cnn.h:100-605)
	int r = model_path ? load_model(ctx, model_path) : HT_OK;
	if (r) return r;
	ctx->cnn_only = model_path == nullptr;
	sync_params(ctx);
	r = ht_alloc_buffers(ctx);
	if (r) return r;
	ctx->ready = true;
	return HT_OK;
}
extern "C" int ht_model_bake(const char *json_path, const char *out_path, int flags)
{
	if (!json_path || !out_path) return HT_ERR_ARG;
	fx_map fx; std::string err;
	if (!ht_build_model(json_path, flags, fx, err)) { fprintf(stderr, "ht_model_bake: %s\n", err.c_str()); return HT_ERR_IO; }
	return fx_save(out_path, fx) ? HT_OK : HT_ERR_IO;
}
// HandTracker::load_config (handtrack.h:822-828) = from_json over visit_fields (:549-581).  The reference's decoder assigns every listed
// field from the file, and a member that is missing (or is not a number) reads as 0 (json.h:104,140,145); a missing file changes nothing.
extern "C" int ht_config_read(const char *jsonfile, ht_params *p, float *segment_scale, float *prev_frame_error)
{
	if (!jsonfile || !p) return HT_ERR_ARG;
	{ FILE *fp = fopen(jsonfile, "rb"); if (!fp) return HT_OK; fclose(fp); }
	std::map<std::string, std::string> num; std::string err;
	if (!ht_json_top_level(jsonfile, num, err)) { fprintf(stderr, "ht_config_read: %s\n", err.c_str()); return HT_ERR_IO; }
	auto F = [&](const char *k) -> float { auto it = num.find(k); return it == num.end() ? 0.0f : strtof(it->second.c_str(), nullptr); };       // istringstream >> float
	auto I = [&](const char *k) -> int { auto it = num.find(k); return it == num.end() ? 0 : (int)strtol(it->second.c_str(), nullptr, 10); };    // istringstream >> int
	p->subsample_voxel = I("subsample_voxel"); p->subsample_size = F("subsample_size");
	if (segment_scale) *segment_scale = F("segment_scale");
	p->full_reset_on_error = F("full_reset_on_error"); p->angles_only = I("angles_only") != 0; p->always_take_cnn = I("always_take_cnn"); p->drangey = F("drangey");
	p->boundary_planes = I("boundary_planes"); p->microforce = F("microforce"); p->mainthreadpasses = I("mainthreadpasses"); p->subsample_fraction = I("subsample_fraction");
	p->min_point_num = I("min_point_num"); p->accum_error_threshold = F("accum_error_threshold"); p->cloudforce_max_point = F("cloudforce_max_point"); p->cloudforce_max_sum = F("cloudforce_max_sum");
	p->steps = I("steps"); p->steps_keypoints = I("steps_keypoints"); p->steps_keyangles = I("steps_keyangles"); p->steps_palmangle = I("steps_palmangle"); p->steps_cloudstart = I("steps_cloudstart");
	if (prev_frame_error) *prev_frame_error = F("prev_frame_error");
	p->physics_iterations = I("physics_iterations"); p->physics_iterations_post = I("physics_iterations_post"); p->physics_use_collision = I("physics_use_collision");
	p->physics_weak_force = F("physics_weak_force"); p->steps_unibody = I("steps_unibody"); p->bone_sum_error_scale = F("bone_sum_error_scale"); p->min_cray_prob = F("min_cray_prob");
	p->unibody_force = F("unibody_force");
	return HT_OK;
}
// HandTracker::scale (handtrack.h:591) = PhysModel::scale on both models (physmodel.h:196-219,304-319): geometry, centres of mass, radii and
// joint anchors times s, inverse inertia over s*s, body positions stretched about the wrist.  The caller keeps segment_scale.
extern "C" int ht_scale(ht_ctx *ctx, float s)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx);
	const int nb = ctx->model.nb, nj = ctx->model.nj;
	const float ss = s * s;
	for (auto &v : ctx->h_verts) { v.x *= s; v.y *= s; v.z *= s; }      // w keeps the vertex index
	for (auto &p : ctx->h_planes) p.w *= s;
	for (int b = 0; b < nb; b++)
	{
		float *c = &ctx->h_bodyc[(size_t)b * HT_BC];
		for (int i = 0; i < 3; i++) c[HT_BC_COM + i] *= s;
		c[HT_BC_RADIUS] *= s; c[HT_BC_RINNER] *= s;
		for (int i = 0; i < 9; i++) c[HT_BC_TINV + i] /= ss;
		float diam2 = 0.0f;      // vertex diameter (used by the exact contact-patch shortcut) of the scaled shape
		const int v0 = ctx->model.vert_off[b], v1 = ctx->model.vert_off[b + 1];
		for (int i = v0; i < v1; i++) for (int j = i + 1; j < v1; j++)
		{
			const float dx = ctx->h_verts[i].x - ctx->h_verts[j].x, dy = ctx->h_verts[i].y - ctx->h_verts[j].y, dz = ctx->h_verts[i].z - ctx->h_verts[j].z;
			const float d2 = dx * dx + dy * dy + dz * dz; if (d2 > diam2) diam2 = d2;
		}
		c[HT_BC_DIAM] = sqrtf(diam2);
	}
	for (int j = 0; j < nj; j++) { float *c = &ctx->h_jointc[(size_t)j * HT_JC]; for (int i = 0; i < 3; i++) { c[HT_JC_P0 + i] *= s; c[HT_JC_P1 + i] *= s; } }
	hipStream_t st = ctx->stream;
	HIPCHK(ctx, hipStreamSynchronize(st));
	HIPCHK(ctx, hipMemcpy(ctx->d_verts_rw, ctx->h_verts.data(), ctx->h_verts.size() * sizeof(float4), hipMemcpyHostToDevice));
	{ const int r = upload_padded_verts(ctx); if (r) return r; }
	HIPCHK(ctx, hipMemcpy(ctx->d_planes_rw, ctx->h_planes.data(), ctx->h_planes.size() * sizeof(float4), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_bodyc_rw, ctx->h_bodyc.data(), ctx->h_bodyc.size() * sizeof(float), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_jointc_rw, ctx->h_jointc.data(), ctx->h_jointc.size() * sizeof(float), hipMemcpyHostToDevice));
	for (int w = 0; w < 2; w++) ht_launch_scale_state(ctx->d_state[w], nb, ctx->B, s, st);
	HIPCHK(ctx, hipStreamSynchronize(st));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
// CNN::Train (cnn.h:558-580), one SGD step per sample in the order given (train-cnn.cpp:156-162 calls it with alpha = 0.001), on the weights
// held by the context; ht_cnn_get_weights reads them back in .cnnb order (CNN::saveb cnn.h:591-593).
extern "C" int ht_cnn_train(ht_ctx *ctx, const float *inputs, const float *targets, int n, float alpha, float *mse_out)
{
	CHECK_READY(ctx);
	if (!ctx->have_weights) { ctx->err = "CNN weights not loaded (ht_cnn_load_weights)"; return HT_ERR_STATE; }
	if (!inputs || !targets || n < 1) return HT_ERR_ARG;
	const size_t na = ht_train_act_floats(), ne = ht_train_err_floats(), np = ht_train_part_floats();
	if (!ctx->d_train) { int r = dev_alloc(ctx, &ctx->d_train, na + ne + np); if (r) return r; }
	float *d_x = nullptr, *d_t = nullptr, *d_mse = nullptr;
	int rc = HT_OK;
	if (hipMalloc((void **)&d_x, (size_t)n * HT_CNN_IN * sizeof(float)) != hipSuccess || hipMalloc((void **)&d_t, (size_t)n * HT_CNN_OUT * sizeof(float)) != hipSuccess ||
	    hipMalloc((void **)&d_mse, (size_t)n * sizeof(float)) != hipSuccess) { ctx->err = "ht_cnn_train: out of device memory"; rc = HT_ERR_HIP; }
	hipStream_t s = ctx->stream;
	if (rc == HT_OK && (hipMemcpyAsync(d_x, inputs, (size_t)n * HT_CNN_IN * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess ||
	                    hipMemcpyAsync(d_t, targets, (size_t)n * HT_CNN_OUT * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)) { ctx->err = "ht_cnn_train: upload failed"; rc = HT_ERR_HIP; }
	if (rc == HT_OK)
	{
		for (int k = 0; k < n; k++)
			ht_launch_train_step(ctx->d_weights, ctx->d_weights + HT_CNNB_COUNT, d_x + (size_t)k * HT_CNN_IN, d_t + (size_t)k * HT_CNN_OUT, alpha, ctx->d_train, ctx->d_train + na, ctx->d_train + na + ne, d_mse + k, s);
		ht_launch_pack_w4(ctx->cnnw.W4, ctx->d_weights + HT_CNNB_COUNT + 16384, s);      // the forward kernels' copy of the last layer follows the trained weights
		if ((mse_out && hipMemcpyAsync(mse_out, d_mse, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) || hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess)
		{ ctx->err = "ht_cnn_train: device error"; rc = HT_ERR_HIP; }
	}
	(void)hipFree(d_x); (void)hipFree(d_t); (void)hipFree(d_mse);
	return rc;
}
extern "C" int ht_cnn_get_weights(ht_ctx *ctx, float *w, size_t n)
{
	CHECK_READY(ctx);
	if (!w) return HT_ERR_ARG;
	if (!ctx->have_weights) { ctx->err = "no weights to read"; return HT_ERR_STATE; }
	if (n != HT_CNNB_COUNT) { ctx->err = "weights: expected HT_CNNB_COUNT fp32 values in .cnnb order"; return HT_ERR_ARG; }
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	HIPCHK(ctx, hipMemcpy(w, ctx->d_weights, n * sizeof(float), hipMemcpyDeviceToHost));
	return HT_OK;
}
// Label synthesis, host only.  GatherHandExpectedCNN (handtrack.h:160-173): 8 landmark heat-maps (ImageFeaturePoints :92-96, RenderHeatMap /
// NormalizeHeatMap misc_image.h:246-272) and 16 one-dimensional maps of the key angles (HandPoseToKeyAngleSet handtrack.h:132-151,
// Render1DHeatMaps misc_image.h:281-295), as bytes scaled by 1/255.  atan2 / asin / acos / exp / pow without std:: are the C double functions there.
static unsigned char gray_of(float x) { float v = x * 255.0f; v = fmin_std(fmax_std(v, 0.0f), 255.0f); return (unsigned char)v; }
extern "C" int ht_expected_cnn_full(const float *pose, const float *cam, float *expected, float *image_points, float *vals_out);
extern "C" int ht_expected_cnn(const float *pose, const float *cam, float *expected) { return ht_expected_cnn_full(pose, cam, expected, nullptr, nullptr); }
// the same with the other members of the Set the reference returns: image_points [8][2] (ImageFeaturePoints handtrack.h:92-96) and vals [16]
extern "C" int ht_expected_cnn_full(const float *pose, const float *cam, float *expected, float *image_points, float *vals_out)
{
	if (!pose || !cam || !expected) return HT_ERR_ARG;
	static const int fbone[8] = { 1, 1, 1, 4, 7, 10, 13, 16 };
	static const float foff[8][3] = { { 0, 0, 0 }, { -0.03f, 0, -0.03f }, { 0.03f, 0, -0.03f }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
	const float fx = cam[0] / 4.0f, fy = cam[1] / 4.0f, px = cam[2] / 4.0f, py = cam[3] / 4.0f;      // hcam = camsub(cam, 4), 16 x 16
	const xf campose = XF(V3(cam[5], cam[6], cam[7]), V4(cam[8], cam[9], cam[10], cam[11])), ci = inverse(campose);
	auto P = [&](int b) { return XF(V3(pose[7 * b], pose[7 * b + 1], pose[7 * b + 2]), V4(pose[7 * b + 3], pose[7 * b + 4], pose[7 * b + 5], pose[7 * b + 6])); };
	unsigned char img[HT_CNN_OUT];
	memset(img, 0, sizeof img);
	for (int k = 0; k < 8; k++)
	{
		const v3 v = apply(ci, apply(P(fbone[k]), V3(foff[k][0], foff[k][1], foff[k][2])));
		const float ux = v.x / v.z * fx + px, uy = v.y / v.z * fy + py;
		if (image_points) { image_points[2 * k] = ux; image_points[2 * k + 1] = uy; }
		unsigned char *h = img + 256 * k;
		const int hx = (int)ux, hy = (int)uy;
		for (int y = (hy - 2 > 0 ? hy - 2 : 0); y < (hy + 3 < 16 ? hy + 3 : 16); y++) for (int x = (hx - 2 > 0 ? hx - 2 : 0); x < (hx + 3 < 16 ? hx + 3 : 16); x++)
		{
			const float dx = ux - (float)x, dy = uy - (float)y;
			h[y * 16 + x] = gray_of(expf(-(dx * dx + dy * dy) / (2.0f * 0.33f)));
		}
		int sum = 0; for (int i = 0; i < 256; i++) sum += h[i];
		if (sum) for (int i = 0; i < 256; i++) h[i] = (unsigned char)(h[i] * 255 / sum);
	}
	float vals[16]; int nv = 0;
	const v4 q1 = P(1).q, palmq = qmul(ci.q, q1);
	vals[nv++] = (float)(atan2((double)qxdir(palmq).x, (double)-qxdir(palmq).z) / (double)(3.14159f * 2.0f) + (double)0.5f);
	vals[nv++] = (float)(asin((double)clamp_std(qzdir(palmq).z, -1.0f, 1.0f)) / (double)3.14159f + (double)0.5f);
	vals[nv++] = (float)(asin((double)clamp_std(qzdir(palmq).x, -1.0f, 1.0f)) / (double)3.14159f + (double)0.5f);
	vals[nv++] = (float)(acos((double)dot(qxdir(q1), qzdir(P(4).q))) / (double)3.14159f);
	for (int b : { 6, 9, 12, 15 }) vals[nv++] = (float)(acos((double)clamp_std(dot(qydir(q1), qydir(P(b).q)), -1.0f, 1.0f)) / (double)3.14159f);
	{ const v3 pz = qzdir(palmq); vals[nv++] = (float)((double)0.5f + atan2((double)-pz.x, (double)-pz.y) / (double)(3.14159f * 2.0f)); }
	while (nv < 16) vals[nv++] = 0.0f;
	if (vals_out) memcpy(vals_out, vals, sizeof vals);
	unsigned char *vm = img + 2048;
	for (int y = 0; y < 16; y++)
	{
		const float v = vals[y] * (float)(16 - 1);
		const int x0 = ((int)v - 2 > 0 ? (int)v - 2 : 0), x1 = ((int)v + 3 < 16 ? (int)v + 3 : 16);
		int sum = 0;
		for (int x = x0; x < x1; x++) { const float d2 = (float)pow((double)((float)x - v), (double)2.0f); sum += vm[y * 16 + x] = gray_of((float)exp((double)(-d2 / (2.0f * 0.5f)))); }
		for (int x = x0; sum && x < x1; x++) vm[y * 16 + x] = (unsigned char)(vm[y * 16 + x] * 255 / sum);
	}
	for (int i = 0; i < HT_CNN_OUT; i++) expected[i] = img[i] / 255.0f;
	return HT_OK;
}
extern "C" int ht_destroy(ht_ctx *ctx)
{
	if (!ctx) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	if (ctx->ready) (void)hipDeviceSynchronize();      // nothing of this context may still run when its buffers go
	if (ctx->comm) (void)ht_comm_destroy(ctx);
	for (void *p : ctx->allocs) (void)hipFree(p);
	if (ctx->h_nreset) (void)hipHostFree(const_cast<unsigned *>(ctx->h_nreset));
	for (auto &kv : ctx->prof) for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
	for (int i = 0; i < 2; i++) { if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]); if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); }
	if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
	if (ctx->ev_lap) (void)hipEventDestroy(ctx->ev_lap);
	if (ctx->ev_job) (void)hipEventDestroy(ctx->ev_job);
	if (ctx->ev_seed) (void)hipEventDestroy(ctx->ev_seed);
	if (ctx->h_job_in) (void)hipHostFree(ctx->h_job_in);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
	return HT_OK;
}
extern "C" const char *ht_last_error(const ht_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" int ht_get_params(const ht_ctx *ctx, ht_params *p) { if (!ctx || !p) return HT_ERR_ARG; *p = ctx->par; return HT_OK; }
// The reference takes any value (load_config writes 0 into every field its file omits, handtrack.h:822-828) and then divides by subsample_fraction
// (physmodel.h:58-64) or loops forever; here a configuration the kernels cannot run is refused with a message instead.
extern "C" int ht_set_params(ht_ctx *ctx, const ht_params *p)
{
	if (!ctx || !p) return HT_ERR_ARG;
	const char *bad = nullptr;
	if (p->subsample_fraction < 1) bad = "subsample_fraction must be >= 1";
	else if (!(p->drangey > 0.1f)) bad = "drangey must exceed the near limit 0.1";
	else if (p->steps < 0 || p->steps_keypoints < 0 || p->steps_keyangles < 0 || p->steps_palmangle < 0 || p->steps_cloudstart < 0 || p->steps_unibody < 0) bad = "step counts must be >= 0";
	else if (p->physics_iterations < 0 || p->physics_iterations_post < 0) bad = "physics iteration counts must be >= 0";
	else if (p->mainthreadpasses < 0) bad = "mainthreadpasses must be >= 0";
	else if (p->min_point_num < 0) bad = "min_point_num must be >= 0";
	else if (p->subsample_voxel && !(p->subsample_size > 0.0f)) bad = "subsample_voxel needs a positive subsample_size (the voxel edge in metres)";
	if (bad) { ctx->err = std::string("ht_set_params: ") + bad; return HT_ERR_ARG; }
	ctx->par = *p; sync_params(ctx);
	return HT_OK;
}
extern "C" int ht_model_info(const ht_ctx *ctx, int *nb, int *nj, int *mb) { if (!ctx) return HT_ERR_ARG; if (nb) *nb = ctx->model.nb; if (nj) *nj = ctx->model.nj; if (mb) *mb = ctx->B; return HT_OK; }


// ------------------------------------------------------------------------------------------------- CNN
extern "C" int ht_cnn_load_weights(ht_ctx *ctx, const float *w, size_t n)
{
	CHECK_READY(ctx);
	if (!w || n != HT_CNNB_COUNT) { ctx->err = "weights: expected HT_CNNB_COUNT fp32 values in .cnnb order"; return HT_ERR_ARG; }
	if (!ctx->d_weights) { int r = dev_alloc(ctx, &ctx->d_weights, (size_t)HT_CNNB_COUNT + 16384 + HT_W4_COUNT); if (r) return r; }
	HIPCHK(ctx, hipMemcpy(ctx->d_weights, w, n * sizeof(float), hipMemcpyHostToDevice));
	// conv2 weights repacked to [k][oc], k = (ky*4+kx)*16 + ic  (reference index: kx + 4*(ky + 4*(ic + 16*oc)), cnn.h:45-47)
	const float *W2 = w + 416;
	std::vector<float> w2p(16384);
	for (int oc = 0; oc < 64; oc++) for (int ic = 0; ic < 16; ic++) for (int ky = 0; ky < 4; ky++) for (int kx = 0; kx < 4; kx++)
		w2p[(size_t)((ky * 4 + kx) * 16 + ic) * 64 + oc] = W2[kx + 4 * (ky + 4 * (ic + 16 * oc))];
	float *d = ctx->d_weights;
	HIPCHK(ctx, hipMemcpy(d + HT_CNNB_COUNT, w2p.data(), 16384 * sizeof(float), hipMemcpyHostToDevice));
	ht_cnn_weights &cw = ctx->cnnw;
	cw.W1 = d; cw.B1 = d + 400; cw.W2p = d + HT_CNNB_COUNT; cw.B2 = d + 416 + 16384; cw.W3 = d + 416 + 16448; cw.B3 = cw.W3 + (size_t)2304 * 2048; cw.W4 = cw.B3 + 2048; cw.B4 = cw.W4 + (size_t)2048 * 2304;
	cw.W4p = d + HT_CNNB_COUNT + 16384;
	ht_launch_pack_w4(cw.W4, d + HT_CNNB_COUNT + 16384, ctx->stream);
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	ctx->have_weights = true;
	return HT_OK;
}
static int cnn_forward(ht_ctx *ctx, const float *d_in, float *d_out, int B, hipStream_t s)
{
	if (!ctx->have_weights) { ctx->err = "CNN weights not loaded (ht_cnn_load_weights)"; return HT_ERR_STATE; }
	ht_prof_scope p0(ctx, "cnn", s, true);
	ht_launch_cnn(ctx->cnnw, d_in, ctx->d_act1, ctx->d_act2, ctx->d_act3, ctx->d_logits, B, s);
	ht_launch_softmax_decode(ctx->d_logits, d_out, nullptr, nullptr, 1, B, s);
	return HT_OK;
}
extern "C" int ht_cnn_eval_dev(ht_ctx *ctx, const float *d_in, float *d_out, int B, void *stream)
{
	CHECK_READY(ctx); CHECK_BATCH(ctx, B);
	if (!d_in || !d_out) return HT_ERR_ARG;
	int r = cnn_forward(ctx, d_in, d_out, B, ht_user_stream(ctx, stream));
	if (r) return r;
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_cnn_eval(ht_ctx *ctx, const float *in, float *out, int B)
{
	CHECK_READY(ctx); CHECK_BATCH(ctx, B);
	if (!in || !out) return HT_ERR_ARG;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_cnn_in, in, (size_t)B * HT_CNN_IN * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	int r = cnn_forward(ctx, ctx->d_cnn_in, ctx->d_cnn_out, B, ctx->stream);
	if (r) return r;
	HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// The same layer list on a 128x128 input (BASELINE configs[4] / SURVEY 8d "config 5 (ii)"): conv5 -> 124, pool -> 62 -> 31, conv4 -> 28, pool -> 14,
// FC 12544 -> 2048 -> 2304, chunked softmax.  A second set of weights and activations beside the 64x64 net's, allocated on first use.
extern "C" int ht_cnn_load_weights_sized(ht_ctx *ctx, int side, const float *w, size_t n)
{
	CHECK_READY(ctx);
	if (side == 64) return ht_cnn_load_weights(ctx, w, n);
	if (side != 128) { ctx->err = "CNN input side must be 64 or 128"; return HT_ERR_ARG; }
	if (!w || n != HT_CNNB128_COUNT) { ctx->err = "weights: expected HT_CNNB128_COUNT fp32 values in .cnnb order"; return HT_ERR_ARG; }
	if (!ctx->d_weights128)
	{
		int r;
		const size_t B = (size_t)ctx->B;
		if ((r = dev_alloc(ctx, &ctx->d_weights128, (size_t)HT_CNNB128_COUNT + 16384 + HT_W4_COUNT)) || (r = dev_alloc(ctx, &ctx->d_in128, B * HT_CNN128_IN)) ||
		    (r = dev_alloc(ctx, &ctx->d_act1_128, B * 16 * 31 * 31)) || (r = dev_alloc(ctx, &ctx->d_act2_128, B * 12544))) return r;
	}
	float *d = ctx->d_weights128;
	HIPCHK(ctx, hipMemcpy(d, w, n * sizeof(float), hipMemcpyHostToDevice));
	const float *W2 = w + 416;
	std::vector<float> w2p(16384);      // conv2 repacked as for the 64x64 net
	for (int oc = 0; oc < 64; oc++) for (int ic = 0; ic < 16; ic++) for (int ky = 0; ky < 4; ky++) for (int kx = 0; kx < 4; kx++)
		w2p[(size_t)((ky * 4 + kx) * 16 + ic) * 64 + oc] = W2[kx + 4 * (ky + 4 * (ic + 16 * oc))];
	HIPCHK(ctx, hipMemcpy(d + HT_CNNB128_COUNT, w2p.data(), 16384 * sizeof(float), hipMemcpyHostToDevice));
	ht_cnn_weights &cw = ctx->cnnw128;
	cw.W1 = d; cw.B1 = d + 400; cw.W2p = d + HT_CNNB128_COUNT; cw.B2 = d + 416 + 16384; cw.W3 = d + 416 + 16448; cw.B3 = cw.W3 + (size_t)12544 * 2048; cw.W4 = cw.B3 + 2048; cw.B4 = cw.W4 + (size_t)2048 * 2304;
	cw.W4p = d + HT_CNNB128_COUNT + 16384;
	ht_launch_pack_w4(cw.W4, d + HT_CNNB128_COUNT + 16384, ctx->stream);
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	ctx->have_weights128 = true;
	return HT_OK;
}
static int cnn128_forward(ht_ctx *ctx, const float *d_in, float *d_out, int B, hipStream_t s)
{
	if (!ctx->have_weights128) { ctx->err = "weights of the 128x128 net not loaded (ht_cnn_load_weights_sized)"; return HT_ERR_STATE; }
	ht_prof_scope p0(ctx, "cnn128", s, true);
	ht_launch_cnn(ctx->cnnw128, d_in, ctx->d_act1_128, ctx->d_act2_128, ctx->d_act3, ctx->d_logits, B, s, 128);
	ht_launch_softmax_decode(ctx->d_logits, d_out, nullptr, nullptr, 1, B, s);
	return HT_OK;
}
extern "C" int ht_cnn_eval_sized_dev(ht_ctx *ctx, int side, const float *d_in, float *d_out, int B, void *stream)
{
	CHECK_READY(ctx); CHECK_BATCH(ctx, B);
	if (side == 64) return ht_cnn_eval_dev(ctx, d_in, d_out, B, stream);
	if (side != 128 || !d_in || !d_out) { ctx->err = "CNN input side must be 64 or 128"; return HT_ERR_ARG; }
	int r = cnn128_forward(ctx, d_in, d_out, B, ht_user_stream(ctx, stream));
	if (r) return r;
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_cnn_eval_sized(ht_ctx *ctx, int side, const float *in, float *out, int B)
{
	CHECK_READY(ctx); CHECK_BATCH(ctx, B);
	if (side == 64) return ht_cnn_eval(ctx, in, out, B);
	if (side != 128 || !in || !out) { ctx->err = "CNN input side must be 64 or 128"; return HT_ERR_ARG; }
	if (!ctx->have_weights128) { ctx->err = "weights of the 128x128 net not loaded (ht_cnn_load_weights_sized)"; return HT_ERR_STATE; }
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_in128, in, (size_t)B * HT_CNN128_IN * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	int r = cnn128_forward(ctx, ctx->d_in128, ctx->d_cnn_out, B, ctx->stream);
	if (r) return r;
	HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// ------------------------------------------------------------------------------------------------- stage: prepare / decode
extern "C" int ht_stage_prepare(ht_ctx *ctx, const uint16_t *depth, const float *cams, int B, float *cnn_in, float *points, int *npoints)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!depth || !cams) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_depth, depth, (size_t)B * 4096 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_cams, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
	ht_launch_prepare(ctx->d_depth, ctx->d_cams, ctx->par.drangey, ctx->par.subsample_fraction, ctx->d_cnn_in, ctx->d_pts, ctx->d_npts, ctx->model.pts_cap, B, s);
	ctx->model.pts_bound = 0;      // stage calls: no assumption about the cloud size
	if (cnn_in) HIPCHK(ctx, hipMemcpyAsync(cnn_in, ctx->d_cnn_in, (size_t)B * HT_CNN_IN * sizeof(float), hipMemcpyDeviceToHost, s));
	if (points) HIPCHK(ctx, hipMemcpy2DAsync(points, HT_MAXPTS * sizeof(float4), ctx->d_pts, (size_t)ctx->model.pts_cap * sizeof(float4), HT_MAXPTS * sizeof(float4), B, hipMemcpyDeviceToHost, s));
	if (npoints) HIPCHK(ctx, hipMemcpyAsync(npoints, ctx->d_npts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_stage_decode(ht_ctx *ctx, const float *cnn_out, const float *cams, int B, float *analysis)
{
	CHECK_READY(ctx); CHECK_BATCH(ctx, B);
	if (!cnn_out || !cams || !analysis) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_cnn_out, cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyHostToDevice, s));
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_cams, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
	ht_launch_softmax_decode(nullptr, ctx->d_cnn_out, ctx->d_cams, ctx->d_analysis, 0, B, s);
	HIPCHK(ctx, hipMemcpyAsync(analysis, ctx->d_analysis, (size_t)B * HT_ANALYSIS * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// ------------------------------------------------------------------------------------------------- profiling hooks
// Each named phase records a pair of HIP events on the stream the kernels run on; pairs are pooled and only resolved in
// ht_profile_read, so enabling the profile adds no host synchronisation to the launch sequence.
ht_prof_scope::ht_prof_scope(ht_ctx *c, const char *name, hipStream_t s, bool minor_phase) : ctx(c), ent(nullptr), stream(s), slot(0)
{
	if (!name || !c->profile || (minor_phase && !c->profile_phases)) return;      // a null name: this scope is not recorded
	auto it = c->prof.find(name);
	if (it == c->prof.end()) { ht_prof_entry e; e.used = 0; e.total_ms = 0; e.launches = 0; it = c->prof.emplace(name, e).first; }
	ht_prof_entry *e = &it->second;
	if (e->used + 2 > e->ev.size())
	{
		hipEvent_t a, b;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
		e->ev.push_back(a); e->ev.push_back(b);
	}
	ent = e; slot = e->used; e->used += 2;
	(void)hipEventRecord(e->ev[slot], s);
}
ht_prof_scope::~ht_prof_scope()
{
	if (!ent) return;
	(void)hipEventRecord(ent->ev[slot + 1], stream);
	ent->launches++;
}
static void prof_resolve(ht_prof_entry &e)
{
	for (size_t i = 0; i + 1 < e.used; i += 2)
	{
		float ms = 0;
		if (hipEventSynchronize(e.ev[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, e.ev[i], e.ev[i + 1]) == hipSuccess) e.total_ms += ms;
	}
	e.used = 0;
}
extern "C" int ht_profile_enable(ht_ctx *ctx, int on) { if (!ctx) return HT_ERR_ARG; ctx->profile = on != 0; ctx->profile_phases = on > 1; return HT_OK; }
extern "C" int ht_profile_read(ht_ctx *ctx, int reset, int max_entries, char *names, int name_stride, float *total_ms, int *launches, int *n_entries)
{
	if (!ctx || !n_entries) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	int k = 0;
	for (auto &kv : ctx->prof)
	{
		ht_prof_entry &e = kv.second;
		prof_resolve(e);
		if (k < max_entries)
		{
			if (names && name_stride > 0) { strncpy(names + (size_t)k * name_stride, kv.first.c_str(), name_stride - 1); names[(size_t)k * name_stride + name_stride - 1] = 0; }
			if (total_ms) total_ms[k] = e.total_ms;
			if (launches) launches[k] = e.launches;
			k++;
		}
		if (reset) { e.total_ms = 0; e.launches = 0; }
	}
	*n_entries = k;
	return HT_OK;
}

// ------------------------------------------------------------------------------------------------- buffers
// The per-point arrays (points, cloud rows, the solver's row records) hold `pts_cap` points per frame.  A context starts with HT_MAXPTS and grows
// when a call brings frames that can carry more (w*h / subsample_fraction), so that no cloud is ever cut: handtrack.h:751 takes any count.
// Growing waits for the context's streams, frees the three arrays and allocates them again (their contents are per-call scratch).
int ht_reserve_points_locked(ht_ctx *ctx, int points)
{
	const int want = (points + 63) & ~63;
	if (ctx->d_pts && want <= ctx->model.pts_cap) return HT_OK;
	const size_t B = (size_t)ctx->B, nb = (size_t)ctx->model.nb, cap = (size_t)want;
	HIPCHK(ctx, ht_sync_all(ctx));
	const bool had_voxel = ctx->d_ptsv != nullptr;
	void *old[5] = { ctx->d_pts, ctx->d_rows, ctx->d_scratch, ctx->d_ptsv, ctx->d_rowbody };
	for (void *o : old) if (o) { for (auto &q : ctx->allocs) if (q == o) { q = ctx->allocs.back(); ctx->allocs.pop_back(); break; } (void)hipFree(o); }
	ctx->d_pts = nullptr; ctx->d_rows = nullptr; ctx->d_scratch = nullptr; ctx->d_ptsv = nullptr; ctx->d_rowbody = nullptr; ctx->model.pts_cap = 0;
	int r;
	if ((r = dev_alloc(ctx, &ctx->d_pts, B * cap))) return r;
	if ((r = dev_alloc(ctx, &ctx->d_rows, B * cap * HT_ROW))) return r;
	if ((r = dev_alloc(ctx, &ctx->d_rowbody, B * cap))) return r;
	if ((r = dev_alloc(ctx, &ctx->d_scratch, B * ht_scratch_rows(cap, nb) * (HT_CREC + 6)))) return r;      // a row record + 1 float for the impulse sum, 1 word for the chain entry (frames that do not fit k_solve's LDS) and 4 floats of couplings per chain entry (ht_quad.hpp: quad_blocks_run); was: k_solve's LDS
	if (had_voxel && (r = dev_alloc(ctx, &ctx->d_ptsv, B * cap))) return r;
	ctx->model.pts_cap = want;
	return HT_OK;
}
extern "C" int ht_reserve_points(ht_ctx *ctx, int points)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx);
	if (points < 1 || points > HT_POINTS_LIMIT) { ctx->err = "ht_reserve_points: between 1 and 76800 points (a 320x240 frame with every pixel in range)"; return HT_ERR_ARG; }
	return ht_reserve_points_locked(ctx, points < HT_MAXPTS ? HT_MAXPTS : points);
}
extern "C" int ht_point_capacity(ht_ctx *ctx, int *points)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx);
	if (!points) return HT_ERR_ARG;
	*points = ctx->model.pts_cap;
	return HT_OK;
}
// the buffer of the solve tables (k_solve_prep's output, ht_solve_shared.hpp): only a context that switches the path on has it
int ht_alloc_solve_tables(ht_ctx *ctx)
{
	if (ctx->d_tables) return HT_OK;
	const size_t B = (size_t)ctx->B;
	int r;
	if ((r = dev_alloc(ctx, &ctx->d_tables, B * TB_WORDS))) return r;
	return HT_OK;
}
int ht_alloc_buffers(ht_ctx *ctx)
{
	const size_t B = (size_t)ctx->B, nb = (size_t)ctx->model.nb;
	int r;
#define A(ptr, n) if ((r = dev_alloc(ctx, &ctx->ptr, (n)))) return r
	A(d_depth, B * 4096); A(d_cams, B * HT_CAM); A(d_cnn_in, B * HT_CNN_IN); A(d_act1, B * 3600); A(d_act2, B * 2304); A(d_act3, B * 2048);
	A(d_logits, B * HT_CNN_OUT); A(d_cnn_out, B * HT_CNN_OUT); A(d_analysis, B * HT_ANALYSIS);
	if (ctx->cnn_only) return HT_OK;
	A(d_npts, B);
	A(d_state[0], B * nb * HT_STATE_STRIDE); A(d_state[1], B * nb * HT_STATE_STRIDE);
	A(d_prev_err, B); A(d_initializing, B); A(d_err_old, B); A(d_err_new, B); A(d_flags, B); A(d_nflags, B);
	A(d_nreset, 2); HIPCHK(ctx, hipMemset(ctx->d_nreset, 0, 2 * sizeof(unsigned)));
	A(d_flist, B); A(d_nflist, 1); HIPCHK(ctx, hipMemset(ctx->d_nflist, 0, sizeof(int)));
	{
		void *h = nullptr;
		HIPCHK(ctx, hipHostMalloc(&h, 2 * sizeof(unsigned), hipHostMallocDefault));
		ctx->h_nreset = static_cast<volatile unsigned *>(h); ctx->h_nreset[0] = ctx->h_nreset[1] = 0;
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
	}
	A(d_nrows, B);
	A(d_chamber, B * 5 * nb * HT_ROW); A(d_nchamber, B); A(d_accepted, B); A(d_chplanes, B * 20); A(d_chon, B);
	if (ht_tuning_int("HT_TABLES", 0) > 0) { if ((r = ht_alloc_solve_tables(ctx))) return r; }      // measurement builds (tools/exp_tables.sh); otherwise on ht_debug_solve_tables(ctx, 1)
	A(d_contacts, B * HT_MAXCONTACT * HT_CONTACT); A(d_ncontacts, B); A(d_epa_ws, ht_contacts_workspace_bytes((int)B)); HIPCHK(ctx, hipMemset(ctx->d_epa_ws, 0, ht_contacts_workspace_bytes((int)B)));
	ctx->cstride = (int)B + 8; A(d_cwork, (size_t)HT_CONTACT_SLOTS * ctx->cstride); A(d_corder, (size_t)HT_CONTACT_SLOTS * ctx->cstride); A(d_porder, B); A(d_swork, (size_t)HT_CONTACT_SLOTS * ctx->cstride); A(d_sorder, (size_t)HT_CONTACT_SLOTS * ctx->cstride);
	if ((r = ht_reserve_points_locked(ctx, HT_MAXPTS))) return r;
	A(d_poses_out, B * nb * HT_POSE); A(d_start, B * nb * HT_POSE);
	A(d_stage, B * nb * HT_STATE_STRIDE);
#undef A
	HIPCHK(ctx, hipMemset(ctx->d_state[0], 0, B * nb * HT_STATE_STRIDE * sizeof(float)));
	HIPCHK(ctx, hipMemset(ctx->d_state[1], 0, B * nb * HT_STATE_STRIDE * sizeof(float)));
	HIPCHK(ctx, hipMemset(ctx->d_prev_err, 0, B * sizeof(float)));
	HIPCHK(ctx, hipMemset(ctx->d_initializing, 0, B * sizeof(int)));
	HIPCHK(ctx, hipMemset(ctx->d_npts, 0, B * sizeof(int)));
	HIPCHK(ctx, hipMemset(ctx->d_nrows, 0, B * sizeof(int)));
	HIPCHK(ctx, hipMemset(ctx->d_ncontacts, 0, B * sizeof(int)));
	HIPCHK(ctx, hipMemset(ctx->d_cwork, 0, (size_t)HT_CONTACT_SLOTS * ctx->cstride * sizeof(int)));      // the work histories: a slot a launch has not written yet ranks equal keys, not garbage
	HIPCHK(ctx, hipMemset(ctx->d_swork, 0, (size_t)HT_CONTACT_SLOTS * ctx->cstride * sizeof(int)));
	return HT_OK;
}
