/*
 * ho_track.c -- CPU restatement of the tracker orchestration (include/handtrack.h), the articulated model and
 * point-cloud constraint generation (include/physmodel.h) and the image helpers on the path (include/misc_image.h).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  The model geometry (vertices, planes, inertia, joints) is not rebuilt here:
 * it is read from the baked fixture tests/golden/model_hand17.htfx that oracle/_ref/ref_harness dumped from the
 * reference's own PhysModel/LoadHandModel (physmodel.h:444-475, handtrack.h:347-366).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ht_oracle.h"

/* from ho_physics.c */
ho_linear ho_along_direction(ho_body *const *B, int rb0, f3 p0, int rb1, f3 p1, f3 axisw, float minforce, float maxforce);
int ho_along_direction_deadzone(ho_body *const *B, int rb0, f3 p0, int rb1, f3 p1, f3 axisw, float radius, f2 forcelimit, ho_linear *out);
int ho_position_nailed(ho_body *const *B, int rb0, f3 p0, int rb1, f3 p1, ho_linear *out);
ho_linear ho_under_plane(ho_body *const *B, int rb, f4 plane, float maxforce);
int ho_angular_range(const ho_physics *ph, ho_body *const *B, int rb0, int rb1, f4 jointframe, f3 lmin, f3 lmax, ho_angular *out);
int ho_angular_drive(const ho_physics *ph, ho_body *const *B, int rb0, int rb1, f4 target, float maxtorque, ho_angular *out);
ho_angular ho_cone_angle(const ho_physics *ph, ho_body *const *B, int rb0, f3 n0, int rb1, f3 n1, float limitangle_degrees);
void ho_sanity_check(ho_model *m);

#define MAXLIN (76800 + 1024)      /* rows of one solve: cloud points (up to a full 320x240 frame, sub-sampling off) + chamber + joints + contacts */
#define MAXANG 256

/* ------------------------------------------------------------------------------------------------ HTFX reader */
typedef struct { char name[48]; uint32_t dtype, ndim, dims[4]; uint64_t nbytes; const unsigned char *data; } fx_entry;
typedef struct { unsigned char *buf; size_t len; fx_entry *e; uint32_t n; } fx_file;
static int fx_open(fx_file *f, const char *path)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) return -1;
	fseek(fp, 0, SEEK_END); f->len = (size_t)ftell(fp); fseek(fp, 0, SEEK_SET);
	f->buf = malloc(f->len);
	if (fread(f->buf, 1, f->len, fp) != f->len) { fclose(fp); return -1; }
	fclose(fp);
	if (f->len < 12 || memcmp(f->buf, "HTFX0001", 8)) return -1;
	memcpy(&f->n, f->buf + 8, 4);
	f->e = calloc(f->n, sizeof(fx_entry));
	size_t off = 12;
	for (uint32_t i = 0; i < f->n; i++)
	{
		memcpy(f->e[i].name, f->buf + off, 48); off += 48;
		memcpy(&f->e[i].dtype, f->buf + off, 4); memcpy(&f->e[i].ndim, f->buf + off + 4, 4); memcpy(f->e[i].dims, f->buf + off + 8, 16); memcpy(&f->e[i].nbytes, f->buf + off + 24, 8); off += 32;
		f->e[i].data = f->buf + off; off += f->e[i].nbytes + ((8 - (f->e[i].nbytes & 7)) & 7);
	}
	return 0;
}
static const fx_entry *fx_find(const fx_file *f, const char *name)      /* optional entry: NULL when absent */
{
	for (uint32_t i = 0; i < f->n; i++) if (!strcmp(f->e[i].name, name)) return &f->e[i];
	return NULL;
}
static const fx_entry *fx_get(const fx_file *f, const char *name)
{
	const fx_entry *e = fx_find(f, name);
	if (!e) fprintf(stderr, "ht_oracle: fixture entry '%s' missing\n", name);
	return e;
}
static void fx_close(fx_file *f) { free(f->buf); free(f->e); }

/* ------------------------------------------------------------------------------------------------ lifecycle */
void ho_default_params(ho_params *p)   /* handtrack.h:523-547 */
{
	p->segment_scale = 0.17f; p->full_reset_on_error = 0.6f; p->angles_only = 0; p->always_take_cnn = 0; p->drangey = 0.7f; p->boundary_planes = 1;
	p->microforce = 1.0f; p->cloudforce_max_point = 15.0f; p->cloudforce_max_sum = 3000.0f; p->mainthreadpasses = 1; p->subsample_fraction = 4;
	p->min_point_num = 400; p->accum_error_threshold = 0.0f; p->min_cray_prob = 0.0f;
	p->steps = 5; p->steps_keypoints = 3; p->steps_keyangles = 2; p->steps_palmangle = 2; p->steps_cloudstart = 1; p->steps_unibody = 3;
	p->subsample_voxel = 0; p->subsample_size = 0.0f;
}
static void world_inertia_init(ho_body *rb) { m33 M = qmat(rb->orientation); rb->Iinv = m33_mul(M, m33_mul(m33_scale(rb->tensorinv_massless, rb->massinv), m33_transpose(M))); }

static int load_model(const fx_file *f, ho_model *m)
{
	const fx_entry *e;
	if (!(e = fx_get(f, "nb"))) return -1; m->nb = *(const int*)e->data;
	if (!(e = fx_get(f, "nj"))) return -1; m->nj = *(const int*)e->data;
	if (m->nb > HO_MAXB || m->nj > HO_MAXJ) return -1;
	const float *bf = (const float*)fx_get(f, "body_f")->data;
	const int *bc = (const int*)fx_get(f, "body_collide")->data;
	const int *ig = (const int*)fx_get(f, "ignore")->data;
	for (int b = 0; b < m->nb; b++)
	{
		ho_body *rb = &m->bodies[b];
		const float *r = bf + 26 * b;
		memset(rb, 0, sizeof *rb);
		rb->mass = r[0]; rb->massinv = r[1]; rb->radius = r[2]; rb->radius_inner = r[3]; rb->damping = r[4]; rb->friction = r[5]; rb->gravscale = r[6];
		rb->com = F3(r[7], r[8], r[9]); rb->position_start = F3(r[10], r[11], r[12]); rb->orientation_start = F4(r[13], r[14], r[15], r[16]);
		rb->tensorinv_massless.x = F3(r[17], r[18], r[19]); rb->tensorinv_massless.y = F3(r[20], r[21], r[22]); rb->tensorinv_massless.z = F3(r[23], r[24], r[25]);
		rb->collide = bc[b];
		rb->position = rb->position_start; rb->orientation = rb->orientation_start;
		world_inertia_init(rb);
		char nm[48];
		snprintf(nm, sizeof nm, "b%d/verts", b); e = fx_get(f, nm); if (!e) return -1;
		rb->shape.nverts = (int)e->dims[0]; rb->shape.verts = malloc(e->nbytes); memcpy(rb->shape.verts, e->data, e->nbytes);
		snprintf(nm, sizeof nm, "b%d/planes", b); e = fx_get(f, nm); if (!e) return -1;
		rb->shape.nplanes = (int)e->dims[0]; rb->shape.planes = malloc(e->nbytes); memcpy(rb->shape.planes, e->data, e->nbytes);
		for (int j = 0; j < m->nb; j++) m->ignore[b][j] = (unsigned char)ig[b * m->nb + j];
	}
	/* HandModelEnhancements' one-time rewrite, handtrack.h:408-416: if bone 2's ignore list has fewer than 10 entries, bone 2 leaves every
	 * collision pair.  It happens on the first call, which precedes every solve with collisions, so it is applied at load.  The list length counts
	 * duplicates in the reference ("ignore_count" of models our builder bakes); files dumped from the reference carry the matrix only. */
	if (m->nb > 2)
	{
		int n2 = 0;
		const fx_entry *ic = fx_find(f, "ignore_count");
		if (ic) n2 = ((const int*)ic->data)[2]; else for (int j = 0; j < m->nb; j++) n2 += m->ignore[2][j] != 0;
		if (n2 < 10) for (int j = 0; j < m->nb; j++) { m->ignore[2][j] = (unsigned char)(j != 2); if (j != 2) m->ignore[j][2] = 1; }
	}
	const int *ji = (const int*)fx_get(f, "joint_i")->data; const float *jf = (const float*)fx_get(f, "joint_f")->data;
	for (int j = 0; j < m->nj; j++)
	{
		ho_joint *jt = &m->joints[j]; const float *r = jf + 16 * j;
		jt->rbi0 = ji[2 * j]; jt->rbi1 = ji[2 * j + 1];
		jt->p0 = F3(r[0], r[1], r[2]); jt->p1 = F3(r[3], r[4], r[5]); jt->rangemin = F3(r[6], r[7], r[8]); jt->rangemax = F3(r[9], r[10], r[11]); jt->jointframe = F4(r[12], r[13], r[14], r[15]);
	}
	return 0;
}
ho_tracker *ho_create(const char *path)
{
	fx_file f;
	if (fx_open(&f, path)) { fprintf(stderr, "ht_oracle: cannot read %s\n", path); return NULL; }
	ho_tracker *t = calloc(1, sizeof *t);
	ho_default_params(&t->par);
	if (load_model(&f, &t->handmodel) || load_model(&f, &t->othermodel)) { fx_close(&f); free(t); return NULL; }
	const float *p = (const float*)fx_get(&f, "physics")->data;
	t->phys.deltaT = p[0]; t->phys.restitution = p[1]; t->phys.gravity = F3(p[2], p[3], p[4]); t->phys.coloumb = p[5]; t->phys.biasfactorjoint = p[6];
	t->phys.biasfactorpositive = p[7]; t->phys.biasfactornegative = p[8]; t->phys.falltime_to_ballistic = p[9]; t->phys.driftmax = p[10]; t->phys.damping = p[11];
	t->phys.iterations = (int)p[12]; t->phys.iterations_post = (int)p[13]; t->phys.use_collision = (int)p[14]; t->phys.weak_force = p[15]; t->phys.bone_sum_error_scale = p[16]; t->phys.unibody_force = p[17];
	{   /* UnibodyFit's cube proxy */
		const fx_entry *e = fx_get(&f, "unibody/verts"); const float *u = (const float*)fx_get(&f, "unibody/f")->data;
		ho_body *ub = &t->unibody_proto; memset(ub, 0, sizeof *ub);
		ub->shape.nverts = (int)e->dims[0]; ub->shape.verts = malloc(e->nbytes); memcpy(ub->shape.verts, e->data, e->nbytes);
		ub->mass = u[0]; ub->massinv = u[1]; ub->radius = u[2]; ub->damping = u[3]; ub->friction = u[4]; ub->gravscale = u[5]; ub->com = F3(u[6], u[7], u[8]);
		ub->tensorinv_massless.x = F3(u[9], u[10], u[11]); ub->tensorinv_massless.y = F3(u[12], u[13], u[14]); ub->tensorinv_massless.z = F3(u[15], u[16], u[17]);
		ub->collide = 3; ub->orientation = F4(0, 0, 0, 1);
	}
	fx_close(&f);
	for (int i = 0; i < HO_NCNN_OUT; i++) t->cnn_output[i] = 0.01f;
	return t;
}
void ho_destroy(ho_tracker *t)
{
	if (!t) return;
	for (int b = 0; b < t->handmodel.nb; b++) { free(t->handmodel.bodies[b].shape.verts); free(t->handmodel.bodies[b].shape.planes); free(t->othermodel.bodies[b].shape.verts); free(t->othermodel.bodies[b].shape.planes); }
	free(t->unibody_proto.shape.verts); free(t->weights); free(t->weights_direct); free(t->cnn_input_direct); free(t);
}
int ho_load_weights(ho_tracker *t, const float *w, size_t n)
{
	if (n != 9458400u) return -1;
	free(t->weights); t->weights = malloc(n * sizeof(float)); memcpy(t->weights, w, n * sizeof(float)); t->nweights = n; return 0;
}
/* the stages of update_cnn_model_threadsafe called directly on a side x side frame with the side-sized net */
int ho_set_direct(ho_tracker *t, int side, const float *w, size_t n)
{
	free(t->weights_direct); free(t->cnn_input_direct); t->weights_direct = NULL; t->cnn_input_direct = NULL; t->direct_side = 0;
	if (side == 0) return 0;
	const size_t feat = (size_t)64 * (((side - 4) / 4 - 3) / 2) * (((side - 4) / 4 - 3) / 2);
	if (side != 128 || !w || n != 400 + 16 + 16384 + 64 + feat * 2048 + 2048 + (size_t)2048 * 2304 + 2304) return -1;
	t->weights_direct = malloc(n * sizeof(float)); memcpy(t->weights_direct, w, n * sizeof(float));
	t->cnn_input_direct = malloc((size_t)side * side * sizeof(float));
	t->direct_side = side;
	return 0;
}
void ho_set_cnn_override(ho_tracker *t, const float *y) { t->cnn_override = y; }
int ho_round_once = 0;
void ho_set_round_once(int on) { ho_round_once = on; }
ho_model *ho_model_ptr(ho_tracker *t, int which) { return which ? &t->othermodel : &t->handmodel; }
int ho_sizeof_tracker(void) { return (int)sizeof(ho_tracker); }
ho_body *ho_body_ptr(ho_model *m, int b) { return &m->bodies[b]; }
void ho_set_state(ho_tracker *t, int which, const float *s)
{
	ho_model *m = ho_model_ptr(t, which);
	for (int b = 0; b < m->nb; b++, s += 13) { ho_body *rb = &m->bodies[b]; rb->position = F3(s[0], s[1], s[2]); rb->orientation = F4(s[3], s[4], s[5], s[6]); rb->linmom = F3(s[7], s[8], s[9]); rb->angmom = F3(s[10], s[11], s[12]); }
}
void ho_set_trace(ho_tracker *t, float *states) { t->trace = states; }
void ho_get_analysis(const ho_tracker *t, float *out84) { memcpy(out84, &t->analysis, 84 * sizeof(float)); }
static void trace_state(ho_tracker *t, ho_model *m, int slot)
{
	if (!t->trace) return;
	float *s = t->trace + (size_t)slot * m->nb * 13;
	for (int b = 0; b < m->nb; b++, s += 13)
	{
		const ho_body *rb = &m->bodies[b];
		s[0] = rb->position.x; s[1] = rb->position.y; s[2] = rb->position.z; s[3] = rb->orientation.x; s[4] = rb->orientation.y; s[5] = rb->orientation.z; s[6] = rb->orientation.w;
		s[7] = rb->linmom.x; s[8] = rb->linmom.y; s[9] = rb->linmom.z; s[10] = rb->angmom.x; s[11] = rb->angmom.y; s[12] = rb->angmom.z;
	}
}
void ho_get_state(ho_tracker *t, int which, float *s)
{
	ho_model *m = ho_model_ptr(t, which);
	for (int b = 0; b < m->nb; b++, s += 13)
	{
		ho_body *rb = &m->bodies[b];
		s[0] = rb->position.x; s[1] = rb->position.y; s[2] = rb->position.z; s[3] = rb->orientation.x; s[4] = rb->orientation.y; s[5] = rb->orientation.z; s[6] = rb->orientation.w;
		s[7] = rb->linmom.x; s[8] = rb->linmom.y; s[9] = rb->linmom.z; s[10] = rb->angmom.x; s[11] = rb->angmom.y; s[12] = rb->angmom.z;
	}
}
void ho_set_pose(ho_tracker *t, int which, const float *s)
{
	ho_model *m = ho_model_ptr(t, which);
	for (int b = 0; b < m->nb; b++, s += 7) { m->bodies[b].position = F3(s[0], s[1], s[2]); m->bodies[b].orientation = F4(s[3], s[4], s[5], s[6]); }
}
void ho_reset_tracker(ho_tracker *t, const float *pose7)
{
	for (int w = 0; w < 2; w++) { ho_set_pose(t, w, pose7); ho_model *m = ho_model_ptr(t, w); for (int b = 0; b < m->nb; b++) m->bodies[b].linmom = m->bodies[b].angmom = F3(0, 0, 0); }
	t->prev_frame_error = 0.0f; t->initializing = 0;
}
void ho_camera_from12(const float *c, int w, int h, ho_camera *cam)
{
	cam->w = w; cam->h = h; cam->focal.x = c[0]; cam->focal.y = c[1]; cam->principal.x = c[2]; cam->principal.y = c[3]; cam->depth_scale = c[4];
	cam->pose = POSE(F3(c[5], c[6], c[7]), F4(c[8], c[9], c[10], c[11]));
}
static void model_ptrs(ho_model *m, ho_body **B) { for (int i = 0; i < m->nb; i++) B[i] = &m->bodies[i]; }

/* ------------------------------------------------------------------------------------------------ image helpers */
static f3 deprojectz(const ho_camera *c, f2 p, float d)   /* misc_image.h:48 */
{
	return scale3(F3((p.x - c->principal.x) / c->focal.x, (p.y - c->principal.y) / c->focal.y, 1.0f), d);
}
static f2 projectz(const ho_camera *c, f3 v) { f2 r = { v.x / v.z * c->focal.x + c->principal.x, v.y / v.z * c->focal.y + c->principal.y }; return r; }   /* misc_image.h:50 */

/* PointCloud misc_image.h:409-417 followed by spatialsubsample physmodel.h:58-64; returns the subsampled count */
int ho_pointcloud(const uint16_t *depth, const ho_camera *cam, float rmin, float rmax, int fraction, f3 *out, int cap, int *n_full)
{
	int k = 0, n = 0;
	for (int y = 0; y < cam->h; y++) for (int x = 0; x < cam->w; x++)
	{
		float d = depth[y * cam->w + x] * cam->depth_scale;
		if (d >= rmin && d < rmax)
		{
			if (k % fraction == 0 && n < cap) { f2 p = { (float)x, (float)y }; out[n++] = deprojectz(cam, p, d); }
			k++;
		}
	}
	if (n_full) *n_full = k;
	return n;
}

/* voxelsubsample<2048> physmodel.h:66-118: points are summed per voxel of `size` metres in an open-addressing table of 2048 buckets (hash = dot of the
 * integer cell with three primes, linear probing), in the order the points come; a table that is full flushes the home bucket of the point that
 * found no room.  Then every bucket with at least `min_count` points gives one point, the mean, in table order.  The reference converts the floor of
 * a NEGATIVE coordinate to unsigned int (undefined in C++); what the compiled reference does, and what this does, is the two's-complement wrap
 * (conversion to a signed integer first).  Returns the number of points written. */
int ho_voxelsubsample(const f3 *pts, int n, float size, int min_count, f3 *out, int cap)
{
	enum { NV = 2048 };
	static struct { int px, py, pz; f3 sum; int cnt; } cand[NV];
	memset(cand, 0, sizeof cand);
	const float ivs = 1.0f / size;
	int m = 0;
	for (int k = 0; k < n; k++)
	{
		const f3 pt = pts[k];
		const int ix = (int)(unsigned int)(long long)floorf(pt.x * ivs), iy = (int)(unsigned int)(long long)floorf(pt.y * ivs), iz = (int)(unsigned int)(long long)floorf(pt.z * ivs);
		const unsigned int hash = (unsigned int)((54851u * (unsigned int)ix + 11909u * (unsigned int)iy) + 24781u * (unsigned int)iz);
		unsigned int i = 0;
		for (; i < NV; i++)
		{
			const unsigned int s = (hash + i) & (NV - 1);
			if (cand[s].cnt == 0 || (cand[s].px == ix && cand[s].py == iy && cand[s].pz == iz))
			{
				cand[s].px = ix; cand[s].py = iy; cand[s].pz = iz;
				cand[s].sum = add3(cand[s].sum, pt); cand[s].cnt++;
				break;
			}
		}
		if (i == NV)
		{
			const unsigned int s = hash & (NV - 1);
			if (m < cap) out[m++] = F3(cand[s].sum.x / (float)cand[s].cnt, cand[s].sum.y / (float)cand[s].cnt, cand[s].sum.z / (float)cand[s].cnt);
			cand[s].cnt = 1; cand[s].px = ix; cand[s].py = iy; cand[s].pz = iz; cand[s].sum = pt;
		}
	}
	for (int s = 0; s < NV; s++) if (cand[s].cnt >= min_count && m < cap) out[m++] = F3(cand[s].sum.x / (float)cand[s].cnt, cand[s].sum.y / (float)cand[s].cnt, cand[s].sum.z / (float)cand[s].cnt);
	return m;
}

/* CNNOutputAnalysis handtrack.h:194-202, 218-241 with ImageFindMax / PeakSubPixel / PeakVolume / Peaks1D misc_image.h:298-399 */
void ho_decode(const float *cnn_output, const ho_camera *hcam, ho_analysis *an)
{
	const int W = 16, H = 16;
	for (int i = 0; i < HO_NLANDMARK; i++)
	{
		const float *base = cnn_output + W * H * i;
		int mxx = 0, mxy = 0;
		for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if (base[y * W + x] > base[mxy * W + mxx]) { mxx = x; mxy = y; }
		float wsum = 0.0f; f2 v = { 0, 0 };
		for (int sy = ho_maxi(0, mxy - 1); sy < ho_mini(H, mxy + 2); sy++) for (int sx = ho_maxi(0, mxx - 1); sx < ho_mini(W, mxx + 2); sx++)
		{
			float w = base[sy * W + sx];
			v.x = v.x + (float)sx * w; v.y = v.y + (float)sy * w;
			wsum += w;
		}
		f2 p; if (wsum == 0) { p.x = (float)mxx; p.y = (float)mxy; } else { p.x = v.x / wsum; p.y = v.y / wsum; }
		an->image_points[i] = p;
		int px = (int)(p.x + 0.5f), py = (int)(p.y + 0.5f);
		float vol = 0.0f;
		for (int sy = ho_maxi(0, py - 1); sy < ho_mini(H, py + 2); sy++) for (int sx = ho_maxi(0, px - 1); sx < ho_mini(W, px + 2); sx++) vol += base[sy * W + sx];
		an->confidence[i] = vol;
		f3 n = normalize3(pose_apply(hcam->pose, deprojectz(hcam, p, 1.0f)));
		an->crays[i] = F4v(n, base[W * mxy + mxx]);
	}
	const float *vptr = cnn_output + W * H * HO_NLANDMARK;
	for (int row = 0; row < HO_NKEYANGLE; row++)
	{
		const float *r = vptr + 16 * row;
		int p = 0;
		for (int x = 1; x < 16; x++) if (r[p] < r[x]) p = x;                    /* std::max_element: first maximum */
		float v = 0.0f, wsum = 0.0f;
		for (int i = ho_maxi(0, p - 1); i < ho_mini(16, p + 2); i++) { float w = r[i]; v += (float)i * w; wsum += w; }
		an->vals[row] = ((wsum == 0) ? (float)p : v / wsum) / (float)(16 - 1);
	}
	/* calc_angles handtrack.h:194-202 (3.1415f literals are the reference's) */
	an->wristroll = an->vals[0] * 3.1415f * 2.0f + 3.1415f / 2.0f;
	an->pitch = (an->vals[1] - 0.5f) * 3.1415f;
	an->tilt = (an->vals[2] - 0.5f) * 3.1415f;
	an->palmq = qmul(normalize4(F4(1.0f, 0, 0, 1.0f)), qmul(quat_axis_angle(F3(-1, 0, 0), an->pitch), quat_axis_angle(F3(0, 0, 1), an->wristroll)));
	for (int i = 0; i < 5; i++) an->finger_clenched[i] = an->vals[3 + i] * 3.1415f;
}

/* ------------------------------------------------------------------------------------------------ closest feature / cloud rows */
static f4 mostabove_world(const ho_body *rb, f3 w)   /* physmodel.h:127-135 */
{
	pose_t P = POSE(rb->position, rb->orientation);
	f3 vl = pose_apply(pose_inverse(P), w);
	f4 vl1 = F4v(vl, 1);
	int best = 0;
	for (int i = 1; i < rb->shape.nplanes; i++) if (dot4(rb->shape.planes[best], vl1) < dot4(rb->shape.planes[i], vl1)) best = i;
	return pose_transform_plane(P, rb->shape.planes[best]);
}
int ho_closest(ho_model *m, f3 v, f4 *plane)   /* physmodel.h:137-162 */
{
	f4 pmin = F4(0, 0, 0, FLT_MAX);
	f4 v1 = F4v(v, 1);
	int rbmin = -1;
	for (int i = 0; i < m->nb; i++)
	{
		const ho_body *rb = &m->bodies[i];
		f3 n = safenormalize3(sub3(v, rb->position));
		f4 p = F4v(n, -dot3(rb->position, n) - rb->radius_inner);
		if (dot4(p, v1) < dot4(pmin, v1)) { pmin = p; rbmin = i; }
	}
	for (int i = 0; i < m->nb; i++)
	{
		const ho_body *rb = &m->bodies[i];
		if (length3(sub3(v, rb->position)) - rb->radius > dot4(pmin, v1)) continue;
		f4 p = mostabove_world(rb, v);
		if (dot4(p, v1) < dot4(pmin, v1)) { pmin = p; rbmin = i; }
	}
	*plane = pmin;
	return rbmin;
}
/* ConvexHitCheck geometric.h:275-302 (posed form) */
static int convex_hit_check(const ho_body *rb, f3 w0, f3 w1, f3 *impact)
{
	pose_t P = POSE(rb->position, rb->orientation), Pi = pose_inverse(P);
	f3 v0 = pose_apply(Pi, w0), v1 = pose_apply(Pi, w1);
	for (int i = 0; i < rb->shape.nplanes; i++)
	{
		f4 plane = rb->shape.planes[i];
		float d0 = dot4(F4v(v0, 1), plane);
		float d1 = dot4(F4v(v1, 1), plane);
		if (d0 >= 0 && d1 >= 0) return 0;
		if (d0 <= 0 && d1 <= 0) continue;
		f3 c = add3(v0, div3(scale3(sub3(v1, v0), d0), (d0 - d1)));
		if (d0 >= 0) v0 = c; else v1 = c;
	}
	*impact = pose_apply(P, v0);
	return 1;
}
ho_linear ho_cloud_constraint(ho_model *m, f3 v, f3 origin)   /* physmodel.h:164-174 */
{
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	f4 p; int rb = ho_closest(m, v, &p);
	const ho_body *b = &m->bodies[rb];
	pose_t Pi = pose_inverse(POSE(b->position, b->orientation));
	f3 impact;
	if (dot3(sub3(v, origin), xyz(p)) > 0 && convex_hit_check(b, origin, v, &impact))
		return ho_along_direction(B, -1, v, rb, pose_apply(Pi, impact), normalize3(sub3(v, origin)), -1.0f, 1.0f);
	return ho_along_direction(B, -1, v, rb, pose_apply(Pi, sub3(v, scale3(xyz(p), dot4(p, F4v(v, 1))))), xyz(p), -1.0f, 1.0f);
}
static int cloud_constraints(ho_model *m, const f3 *pts, int n, int stride, f3 origin, ho_linear *out)
{
	int k = 0;
	for (int i = 0; i < n; i += stride) out[k++] = ho_cloud_constraint(m, pts[i], origin);
	return k;
}
/* containing_plane / cloud_chamber physmodel.h:183-193, 486-496 with the outdirs of handtrack.h:776 */
static f4 containing_plane(const f3 *pts, int n, f3 outdir, f3 origin, f3 viewdir)
{
	f3 best = sub3(viewdir, outdir);
	best = add3(best, origin);
	f3 tangent = cross3(best, outdir);
	for (int i = 0; i < n; i++)
		if (dot3(cross3(sub3(best, origin), sub3(pts[i], origin)), tangent) > 0) best = pts[i];
	f3 nn = normalize3(cross3(tangent, best));
	return F4v(nn, -dot3(nn, origin));
}
int ho_cloud_chamber(ho_model *m, const f3 *pts, int n, ho_linear *out, float maxforce)
{
	static const float od[5][3] = { { -1, -0.25f, 0 }, { -1, -1, 0 }, { 0, -1, 0 }, { 1, -1, 0 }, { 1, -0.25f, 0 } };
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	int k = 0;
	for (int d = 0; d < 5; d++)
	{
		f4 cplane = containing_plane(pts, n, F3(od[d][0], od[d][1], od[d][2]), F3(0, 0, 0), F3(0, 0, 1));
		for (int b = 0; b < m->nb; b++) out[k++] = ho_under_plane(B, b, cplane, maxforce);
	}
	return k;
}

/* ------------------------------------------------------------------------------------------------ joints */
int ho_joint_linears(ho_model *m, ho_linear *out)   /* physmodel.h:328-334 */
{
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	int k = 0;
	for (int j = 0; j < m->nj; j++)
	{
		const ho_joint *jt = &m->joints[j];
		k += ho_position_nailed(B, jt->rbi0, sub3(jt->p0, m->bodies[jt->rbi0].com), jt->rbi1, sub3(jt->p1, m->bodies[jt->rbi1].com), out + k);
	}
	return k;
}
int ho_joint_angulars(ho_tracker *t, ho_model *m, ho_angular *out)   /* physmodel.h:321-327 */
{
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	int k = 0;
	for (int j = 0; j < m->nj; j++)
	{
		const ho_joint *jt = &m->joints[j];
		k += ho_angular_range(&t->phys, B, jt->rbi0, jt->rbi1, jt->jointframe, jt->rangemin, jt->rangemax, out + k);
	}
	return k;
}
/* HandModelEnhancements handtrack.h:406-441.  acos()/cos() there are the C double overloads.
 * The ignore-list rewrite at :408-416 is applied when the model is loaded (see there). */
void ho_enhancements(ho_tracker *t, ho_model *m, ho_angular *ang, int *nang, int tiepinkyringmid, f3 palmxdir, f3 armdir, int fingerhold)
{
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	(void)palmxdir;
	for (int b = 7; b < 17; b += 3)
	{
		float c = ho_clampf(dot3(qzdir(m->bodies[b - 2].orientation), qzdir(m->bodies[b - 1].orientation)), 0.0f, 1.0f);
		float a = (float)(acos((double)c) * (double)180.0f / (double)3.14159f / (double)2.0f);
		m->joints[b - 1].rangemax.x = a; m->joints[b - 1].rangemin.x = a;
	}
	if (tiepinkyringmid)
	{
		static const int bs[4] = { 15, 14, 12, 11 };
		for (int i = 0; i < 4; i++) ang[(*nang)++] = ho_cone_angle(&t->phys, B, bs[i], F3(0, 1, 0), bs[i] - 3, F3(0, 1, 0), 10.0f);
	}
	if (!(armdir.x == 0 && armdir.y == 0 && armdir.z == 0))
		ang[(*nang)++] = ho_cone_angle(&t->phys, B, -1, armdir, 0, F3(0, 0, 1), 70.0f);
	if (fingerhold & 1) ang[(*nang)++] = ho_cone_angle(&t->phys, B, 1, F3(-1, 0, 0), 4, F3(0, 0, 1), 10.0f);
	for (int finger = 1; finger <= 4; finger++)
		if (fingerhold & (1 << finger)) ang[(*nang)++] = ho_cone_angle(&t->phys, B, 1, F3(0, 0, -1), 3 + finger * 3, F3(0, 0, 1), 10.0f);
	static const struct { int bone; float r0, r1; } kl[4] = { { 14, -30.0f, 10.0f }, { 11, -10.0f, 10.0f }, { 8, -10.0f, 10.0f }, { 5, -10.0f, 20.0f } };
	for (int i = 0; i < 4; i++)
	{
		int up = (double)dot3(qydir(m->bodies[1].orientation), qydir(m->bodies[kl[i].bone].orientation)) > cos((double)(40.0f * 3.14f / 180.0f));
		m->joints[kl[i].bone - 1].rangemin.y = up ? kl[i].r0 : -0.0f;
		m->joints[kl[i].bone - 1].rangemax.y = up ? kl[i].r1 : 0.0f;
	}
}
/* CNNOutputAnalysis::ApplyAngles handtrack.h:203-216; cos()/sin() there are the C double overloads, narrowed to float */
int ho_apply_angles(ho_tracker *t, ho_model *m, const ho_analysis *an, pose_t camera_pose, float drive_force, float coneangle, ho_angular *out)
{
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	int k = ho_angular_drive(&t->phys, B, -1, 1, qmul(camera_pose.orientation, an->palmq), drive_force, out);
	float thumbangle = an->finger_clenched[0];
	out[k++] = ho_cone_angle(&t->phys, B, 1, F3((float)cos((double)thumbangle), 0, (float)sin((double)thumbangle)), 4, F3(0, 0, 1), coneangle);
	for (int finger = 1; finger <= 4; finger++)
	{
		float a = an->finger_clenched[finger];
		out[k++] = ho_cone_angle(&t->phys, B, 1, F3(0, (float)(-sin((double)a)), (float)cos((double)a)), 3 + finger * 3, F3(0, 0, 1), coneangle);
		f4 jf = m->joints[1 + finger * 3].jointframe;
		f3 inner = F3(0, (float)(-sin((double)(a / 2.0f))), (float)cos((double)(a / 2.0f)));
		out[k++] = ho_cone_angle(&t->phys, B, 1, qrot(jf, qrot(jf, inner)), 2 + finger * 3, F3(0, 0, 1), coneangle);
	}
	return k;
}

/* ------------------------------------------------------------------------------------------------ fit steps */
/* PhysModel::FitPointCloud physmodel.h:345-356 */
void ho_fit_pointcloud(ho_tracker *t, ho_model *m, const f3 *pts, int n, const ho_linear *lin_in, int nlin_in, const ho_angular *ang_in, int nang_in, float microforce)
{
	ho_linear *lin = malloc(sizeof(ho_linear) * MAXLIN); ho_angular ang[MAXANG];
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	int nl = 0, na = 0;
	for (int i = 0; i < nlin_in; i++) lin[nl++] = lin_in[i];
	for (int i = 0; i < nang_in; i++) ang[na++] = ang_in[i];
	int c0 = nl;
	nl += cloud_constraints(m, pts, n, 1, F3(0, 0, 0), lin + nl);
	for (int i = c0; i < nl; i++)
	{
		float k = (lin[i].rb1 == 0 || lin[i].rb1 == 1 || lin[i].rb1 == 2) ? t->phys.weak_force : 1.0f;
		lin[i].forcelimit.x = -1.0f * k * microforce; lin[i].forcelimit.y = 1.0f * k * microforce;
	}
	nl += ho_joint_linears(m, lin + nl);
	na += ho_joint_angulars(t, m, ang + na);
	ho_physics_update(t, B, m->nb, m, lin, nl, MAXLIN, ang, na);
	ho_sanity_check(m);
	free(lin);
}
/* feature points handtrack.h:77-81 */
static const struct { int bone; float off[3]; } FEATURE[8] = { { 1, { 0, 0, 0 } }, { 1, { -0.03f, 0, -0.03f } }, { 1, { 0.03f, 0, -0.03f } }, { 4, { 0, 0, 0 } }, { 7, { 0, 0, 0 } }, { 10, { 0, 0, 0 } }, { 13, { 0, 0, 0 } }, { 16, { 0, 0, 0 } } };

/* HandTracker::MultiStepSim handtrack.h:642-690 */
void ho_multistep(ho_tracker *t, ho_model *m, const ho_analysis *an, const f3 *vpts, int n, pose_t camera_pose)
{
	const ho_params *P = &t->par;
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	ho_linear *lin = malloc(sizeof(ho_linear) * MAXLIN); ho_angular ang[MAXANG];
	ho_sanity_check(m);
	float cloudforce = ho_minf(P->cloudforce_max_point, P->cloudforce_max_sum / (float)n);
	trace_state(t, m, 0);
	for (int s = 0; s < P->steps; s++)
	{
		int nl = 0, na = 0;
		if (s < P->steps_keyangles || P->angles_only)
			na += ho_apply_angles(t, m, an, camera_pose, s < P->steps_palmangle ? 10000.0f : 0.0f, 10.0f, ang + na);
		if (s < P->steps_keypoints && !P->angles_only)
		{
			for (int i = (P->steps_keyangles ? 3 : 0); i < 8; i++) if (i >= 3 && an->finger_clenched[i - 3] < 3.14f / 2.0f && an->crays[i].w >= P->min_cray_prob)
			{
				f4 q = quat_from_to(F3(0, 0, 1), xyz(an->crays[i]));
				f3 off = F3(FEATURE[i].off[0], FEATURE[i].off[1], FEATURE[i].off[2]);
				f2 fl = { -100000.0f, 100000.0f };
				nl += ho_along_direction_deadzone(B, -1, camera_pose.position, FEATURE[i].bone, off, qxdir(q), 0.01f, fl, lin + nl);
				nl += ho_along_direction_deadzone(B, -1, camera_pose.position, FEATURE[i].bone, off, qydir(q), 0.01f, fl, lin + nl);
			}
		}
		if (s >= P->steps_cloudstart && n && cloudforce > 0.0f && !P->angles_only)
		{
			int c0 = nl;
			nl += cloud_constraints(m, vpts, n, 4, camera_pose.position, lin + nl);   /* takesubsample(vpts): every 4th */
			for (int i = c0; i < nl; i++)
			{
				float k = (lin[i].rb1 == 0) ? 0.1f : 1.0f;
				lin[i].forcelimit.x = -cloudforce * k; lin[i].forcelimit.y = cloudforce * k;
			}
		}
		ho_enhancements(t, m, ang, &na, 0, qrot(camera_pose.orientation, F3(-1, 0, 0)), qrot(camera_pose.orientation, F3(0, -1, 0)), 0);
		ho_fit_pointcloud(t, m, NULL, 0, lin, nl, ang, na, 1.0f);
		for (int b = 0; b < m->nb; b++) m->bodies[b].angmom = m->bodies[b].linmom = F3(0, 0, 0);
		trace_state(t, m, s + 1);
	}
	ho_sanity_check(m);
	free(lin);
}

/* FitError handtrack.h:371-399 */
float ho_fit_error(ho_tracker *t, ho_model *m, const f3 *pts, int n, const uint16_t *depth, const ho_camera *cam)
{
	float pointerror[HO_MAXB];
	for (int b = 0; b < m->nb; b++) pointerror[b] = 0.0f;
	for (int i = 0; i < n; i++)
	{
		f4 p; int b = ho_closest(m, pts[i], &p);
		pointerror[b] = ho_maxf(pointerror[b], dot4(p, F4v(pts[i], 1.0f)));
	}
	float point_error_sum = 0.0f;
	for (int b = 0; b < m->nb; b++) point_error_sum += pointerror[b];
	float bone_error_sum = 0;
	pose_t ci = pose_inverse(cam->pose);
	for (int b = 0; b < m->nb; b++)
	{
		f3 position = pose_apply(ci, m->bodies[b].position);
		f2 pf = projectz(cam, position);
		int px = (int)pf.x, py = (int)pf.y;
		if (!(px >= 0 && px <= cam->w - 1 && py >= 0 && py <= cam->h - 1)) continue;
		float bone_error = depth[py * cam->w + px] * cam->depth_scale - position.z;
		bone_error_sum += ho_clampf(bone_error, 0.0f, 0.01f);
	}
	return point_error_sum + bone_error_sum * t->phys.bone_sum_error_scale;
}

/* FixPositions physmodel.h:404-408 */
static void fix_positions(ho_model *m)
{
	for (int j = 0; j < m->nj; j++)
	{
		const ho_joint *jt = &m->joints[j];
		ho_body *r0 = &m->bodies[jt->rbi0], *r1 = &m->bodies[jt->rbi1];
		pose_t u0 = POSE(pose_apply(POSE(r0->position, r0->orientation), neg3(r0->com)), r0->orientation);
		pose_t u1 = POSE(pose_apply(POSE(r1->position, r1->orientation), neg3(r1->com)), r1->orientation);
		r1->position = add3(r1->position, sub3(pose_apply(u0, jt->p0), pose_apply(u1, jt->p1)));
	}
}
/* PoseFromScratch handtrack.h:480-506 */
void ho_pose_from_scratch(ho_tracker *t, ho_model *m, const f3 *pts, int n, const ho_analysis *an, pose_t camera_pose)
{
	(void)t;
	f4 cs = add4(add4(an->crays[0], an->crays[1]), an->crays[2]);
	f3 palmray = normalize3(xyz(cs));
	f3 pcom = F3(0, 0, 0); float wsum = 0.00000000001f;
	for (int i = 0; i < n; i++)
	{
		f3 c = cross3(pts[i], palmray);
		float w = 1.0f / (0.000001f + dot3(c, c));
		pcom = add3(pcom, scale3(pts[i], w)); wsum += w;
	}
	pcom = div3(pcom, wsum);
	for (int b = 0; b < m->nb; b++) { ho_body *rb = &m->bodies[b]; rb->position = rb->position_start; rb->orientation = rb->orientation_start; rb->linmom = rb->angmom = F3(0, 0, 0); }
	pose_t p1 = POSE(pcom, qmul(camera_pose.orientation, an->palmq));
	pose_t dp = pose_mul(p1, pose_inverse(POSE(m->bodies[1].position, m->bodies[1].orientation)));
	for (int b = 0; b < m->nb; b++) { pose_t np = pose_mul(dp, POSE(m->bodies[b].position, m->bodies[b].orientation)); m->bodies[b].position = np.position; m->bodies[b].orientation = np.orientation; }
	for (int finger = 1; finger <= 4; finger++)
	{
		float a = an->finger_clenched[finger];
		f4 jf = m->joints[1 + finger * 3].jointframe;
		m->bodies[2 + finger * 3].orientation = qmul(jf, qmul(m->bodies[2 + finger * 3].orientation, quat_axis_angle(F3(1, 0, 0), a / 2.0f)));
		m->bodies[3 + finger * 3].orientation = qmul(jf, qmul(m->bodies[3 + finger * 3].orientation, quat_axis_angle(F3(1, 0, 0), a)));
		m->bodies[4 + finger * 3].orientation = qmul(jf, qmul(m->bodies[4 + finger * 3].orientation, quat_axis_angle(F3(1, 0, 0), a * 1.25f)));
	}
	fix_positions(m);
}
/* UnibodyFit handtrack.h:451-470 */
void ho_unibody_fit(ho_tracker *t, ho_model *m, const f3 *pts, int n, f3 camera_position)
{
	ho_linear *lin = malloc(sizeof(ho_linear) * MAXLIN);
	int nl = cloud_constraints(m, pts, n, 4, camera_position, lin);
	ho_body ub = t->unibody_proto;
	ub.position = add3(m->bodies[1].position, ub.com);            /* RigidBody ctor: position += com (physics.h:157) */
	ub.orientation = m->bodies[1].orientation;
	ub.linmom = ub.angmom = F3(0, 0, 0);
	pose_t ubi = pose_inverse(POSE(ub.position, ub.orientation));
	for (int i = 0; i < nl; i++)
	{
		const ho_body *rb1 = &m->bodies[lin[i].rb1];
		lin[i].position1 = pose_apply(ubi, pose_apply(POSE(rb1->position, rb1->orientation), lin[i].position1));
		lin[i].rb1 = 0;
		lin[i].forcelimit.x *= t->phys.unibody_force; lin[i].forcelimit.y *= t->phys.unibody_force;
	}
	ho_sanity_check(m);
	ho_body *UB[1] = { &ub };
	ho_physics_update(t, UB, 1, NULL, lin, nl, MAXLIN, NULL, 0);
	pose_t dp = pose_mul(POSE(ub.position, ub.orientation), pose_inverse(POSE(m->bodies[1].position, m->bodies[1].orientation)));
	for (int b = 0; b < m->nb; b++) { pose_t np = pose_mul(dp, POSE(m->bodies[b].position, m->bodies[b].orientation)); m->bodies[b].position = np.position; m->bodies[b].orientation = np.orientation; }
	ho_sanity_check(m);
	free(lin);
}

/* ------------------------------------------------------------------------------------------------ tracker */
/* update_cnn_model(_threadsafe) handtrack.h:693-741 for a 64x64 tile (HandSegmentVR is the identity, :283-284) */
int ho_update_cnn_model(ho_tracker *t, const uint16_t *depth, const ho_camera *cam, float *pose_out7)
{
	const ho_params *P = &t->par;
	float drx = 0.1f, dry = P->drangey;
	/* handtrack.h:697-698: segment = HandSegmentVR(dimage, 0xF, drange, segment_scale); a 64x64 image is its own segment (:283-284).
	 * The CNN sees the segment; points and FitError keep using the full image; segment.cam.pose goes to the pose-driven stages. */
	uint16_t tile[64 * 64];
	ho_camera scam = *cam;
	const uint16_t *seg = depth;
	const int direct = t->direct_side && cam->w == t->direct_side && cam->h == t->direct_side;      /* the frame is its own segment, as a 64x64 one is for the stock net */
	if (!direct && (cam->w != 64 || cam->h != 64))
	{
		const float c12[12] = { cam->focal.x, cam->focal.y, cam->principal.x, cam->principal.y, cam->depth_scale, cam->pose.position.x, cam->pose.position.y, cam->pose.position.z,
		                        cam->pose.orientation.x, cam->pose.orientation.y, cam->pose.orientation.z, cam->pose.orientation.w };
		float o12[12];
		ho_segment_vr(depth, cam->w, cam->h, c12, 0xF, drx, dry, P->segment_scale, tile, o12, NULL, NULL);
		ho_camera_from12(o12, 64, 64, &scam);
		seg = tile;
	}
	const int sub = direct ? cam->w / 16 : 4;
	ho_camera hcam = scam;   /* camsub(cam,4) misc_image.h:60 (direct: camsub(cam, side/16), 16x16 heat-maps again) */
	hcam.w = scam.w / sub; hcam.h = scam.h / sub; hcam.focal.x = scam.focal.x / (float)sub; hcam.focal.y = scam.focal.y / (float)sub; hcam.principal.x = scam.principal.x / (float)sub; hcam.principal.y = scam.principal.y / (float)sub;
	if (direct)
	{
		ho_cnn_input(seg, cam->w * cam->h, scam.depth_scale, drx, dry, t->cnn_input_direct);
		ho_cnn_eval_sized(t->weights_direct, t->direct_side, t->cnn_input_direct, t->cnn_output, NULL);
	}
	else
	{
		ho_cnn_input(seg, 64 * 64, scam.depth_scale, drx, dry, t->cnn_input);
		ho_cnn_eval(t->weights, t->cnn_input, t->cnn_output, NULL);
	}
	if (t->cnn_override) memcpy(t->cnn_output, t->cnn_override, sizeof(float) * HO_NCNN_OUT);
	ho_decode(t->cnn_output, &hcam, &t->analysis);
	f3 *vpts = malloc(sizeof(f3) * cam->w * cam->h);
	int n = ho_pointcloud(depth, cam, drx, dry, P->subsample_fraction, vpts, cam->w * cam->h, NULL);
	float olderror = ho_fit_error(t, &t->handmodel, vpts, n, depth, cam);
	if (P->angles_only || olderror > P->full_reset_on_error)
	{
		ho_pose_from_scratch(t, &t->othermodel, vpts, n, &t->analysis, scam.pose);
		for (int i = 0; i < P->steps_unibody; i++) ho_unibody_fit(t, &t->othermodel, vpts, n, scam.pose.position);
	}
	ho_multistep(t, &t->othermodel, &t->analysis, vpts, n, scam.pose);
	float newerror = ho_fit_error(t, &t->othermodel, vpts, n, depth, cam);
	if (newerror > olderror) t->prev_frame_error = 0.0f; else t->prev_frame_error += olderror - newerror;
	int np = 0;
	if ((n > P->min_point_num && t->initializing) || P->always_take_cnn || P->angles_only || t->prev_frame_error > P->accum_error_threshold)
	{
		np = t->othermodel.nb;
		for (int b = 0; b < np; b++)
		{
			const ho_body *rb = &t->othermodel.bodies[b];
			float *o = pose_out7 + 7 * b;
			o[0] = rb->position.x; o[1] = rb->position.y; o[2] = rb->position.z; o[3] = rb->orientation.x; o[4] = rb->orientation.y; o[5] = rb->orientation.z; o[6] = rb->orientation.w;
		}
	}
	if (t->prev_frame_error > P->accum_error_threshold) t->prev_frame_error = 0.0f;
	t->initializing = ho_maxi(t->initializing - 1, 0);
	t->last_accept = np;
	free(vpts);
	return np;
}
void ho_get_flags(const ho_tracker *t, float *prev_frame_error, int *initializing, int *last_npoints)
{
	if (prev_frame_error) *prev_frame_error = t->prev_frame_error;
	if (initializing) *initializing = t->initializing;
	if (last_npoints) *last_npoints = t->last_npoints;
}
/* HandTracker::update handtrack.h:748-785 with the background job run synchronously every frame */
void ho_update(ho_tracker *t, const uint16_t *depth, const ho_camera *cam, float *pose_user_out7)
{
	const ho_params *P = &t->par;
	f3 *points = malloc(sizeof(f3) * cam->w * cam->h);
	int n;
	if (P->subsample_voxel)      /* takesubsample physmodel.h:120-126: the voxel branch takes ALL in-range points; subsample_fraction is its minimum count */
	{
		f3 *all = malloc(sizeof(f3) * cam->w * cam->h);
		const int na = ho_pointcloud(depth, cam, 0.1f, P->drangey, 1, all, cam->w * cam->h, NULL);
		n = ho_voxelsubsample(all, na, P->subsample_size, P->subsample_fraction, points, cam->w * cam->h);
		free(all);
	}
	else n = ho_pointcloud(depth, cam, 0.1f, P->drangey, P->subsample_fraction, points, cam->w * cam->h, NULL);
	t->last_npoints = n;
	for (int b = 0; b < t->handmodel.nb; b++) { t->othermodel.bodies[b].position = t->handmodel.bodies[b].position; t->othermodel.bodies[b].orientation = t->handmodel.bodies[b].orientation; }
	float pose[HO_MAXB * 7];
	int np = ho_update_cnn_model(t, depth, cam, pose);
	if (np) ho_set_pose(t, 0, pose);
	trace_state(t, &t->handmodel, P->steps + 1);
	for (int i = 0; !P->angles_only && i < P->mainthreadpasses; i++)
	{
		ho_linear *lin = malloc(sizeof(ho_linear) * 256); ho_angular ang[16];
		int nl = 0, na = 0;
		ho_enhancements(t, &t->handmodel, ang, &na, 0, F3(0, 0, 0), F3(0, 0, 0), 0);
		if (n > P->min_point_num && P->boundary_planes) nl += ho_cloud_chamber(&t->handmodel, points, n, lin, 10.0f);
		ho_fit_pointcloud(t, &t->handmodel, points, n, lin, nl, ang, na, P->microforce);
		free(lin);
		trace_state(t, &t->handmodel, P->steps + 2 + i);
	}
	if (n < P->min_point_num) t->initializing = 50;
	for (int b = 0; b < t->handmodel.nb; b++)   /* GetPoseUser physmodel.h:434, physics.h:142 */
	{
		const ho_body *rb = &t->handmodel.bodies[b];
		f3 pu = pose_apply(POSE(rb->position, rb->orientation), neg3(rb->com));
		float *o = pose_user_out7 + 7 * b;
		o[0] = pu.x; o[1] = pu.y; o[2] = pu.z; o[3] = rb->orientation.x; o[4] = rb->orientation.y; o[5] = rb->orientation.z; o[6] = rb->orientation.w;
	}
	free(points);
}

/* PhysModel::scale (physmodel.h:196-219,304-319) on both models, as HandTracker::scale does (handtrack.h:591) */
static void ho_scale_model(ho_model *m, float s)
{
	for (int b = 0; b < m->nb; b++)
	{
		ho_body *rb = &m->bodies[b];
		for (int i = 0; i < rb->shape.nverts; i++) rb->shape.verts[i] = scale3(rb->shape.verts[i], s);
		for (int i = 0; i < rb->shape.nplanes; i++) rb->shape.planes[i].w *= s;
		rb->com = scale3(rb->com, s);
		rb->radius *= s; rb->radius_inner *= s;
		const float ss = s * s;
		rb->tensorinv_massless.x = div3(rb->tensorinv_massless.x, ss); rb->tensorinv_massless.y = div3(rb->tensorinv_massless.y, ss); rb->tensorinv_massless.z = div3(rb->tensorinv_massless.z, ss);
		rb->Iinv.x = div3(rb->Iinv.x, ss); rb->Iinv.y = div3(rb->Iinv.y, ss); rb->Iinv.z = div3(rb->Iinv.z, ss);
	}
	for (int b = 0; b < m->nb; b++)
		m->bodies[b].position = add3(m->bodies[0].position, scale3(sub3(m->bodies[b].position, m->bodies[0].position), s));
	for (int j = 0; j < m->nj; j++) { m->joints[j].p0 = scale3(m->joints[j].p0, s); m->joints[j].p1 = scale3(m->joints[j].p1, s); }
}
void ho_scale(ho_tracker *t, float s) { ho_scale_model(&t->handmodel, s); ho_scale_model(&t->othermodel, s); }

/* PhysModel::RelativeAngularConstraints(refpose, filter) physmodel.h:423-432 with slowfit's filter (handtrack.h:799):
 * joint j passes when (j != 0 && hold == 2) || j > 3.  Uses the joint ranges HandModelEnhancements just wrote. */
static pose_t body_pose(const ho_body *b) { return POSE(b->position, b->orientation); }
static int relative_angular(ho_tracker *t, ho_model *m, const float *refpose7, int hold, ho_angular *out)
{
	int k = 0;
	for (int j = 0; j < m->nj; j++)
	{
		if (!((j != 0 && hold == 2) || j > 3)) continue;
		const ho_joint *jt = &m->joints[j];
		const float *r0 = refpose7 + 7 * jt->rbi0, *r1 = refpose7 + 7 * jt->rbi1;
		pose_t ref0 = POSE(F3(r0[0], r0[1], r0[2]), F4(r0[3], r0[4], r0[5], r0[6])), ref1 = POSE(F3(r1[0], r1[1], r1[2]), F4(r1[3], r1[4], r1[5], r1[6]));
		/* dq = (ref0^-1 * ref1)^-1 * rb0.pose^-1 * rb1.pose, evaluated left to right */
		pose_t dq = pose_mul(pose_mul(pose_inverse(pose_mul(pose_inverse(ref0), ref1)), pose_inverse(body_pose(&m->bodies[jt->rbi0]))), body_pose(&m->bodies[jt->rbi1]));
		m33 R = qmat(m->bodies[jt->rbi0].orientation);
		for (int a = 0; a < 3; a++) if (f3_get(jt->rangemin, a) != f3_get(jt->rangemax, a))
		{
			ho_angular *o = &out[k++];
			o->rb0 = jt->rbi0; o->rb1 = jt->rbi1; o->axis = a == 0 ? R.x : a == 1 ? R.y : R.z; o->torque = 0;
			o->targetspin = -f4_get(dq.orientation, a) * 2.0f / t->phys.deltaT; o->mintorque = -FLT_MAX; o->maxtorque = FLT_MAX;
		}
	}
	return k;
}
/* HandTracker::slowfit handtrack.h:786-821 on handmodel; selectrb < 0: none; crays: ncray x (dir3, weight) */
void ho_slowfit(ho_tracker *t, const f3 *points, int n, int hold, const float *refpose7, int steps_, int selectrb, f3 spoint, f3 rbpoint, const float *crays4, int ncray)
{
	ho_model *m = &t->handmodel;
	ho_body *B[HO_MAXB]; model_ptrs(m, B);
	ho_linear *lin = malloc(sizeof(ho_linear) * MAXLIN); ho_angular ang[MAXANG];
	for (int st = 0; st < steps_; st++)
	{
		int nl = 0, na = 0;
		ho_enhancements(t, m, ang, &na, 0, F3(0, 0, 0), F3(0, 0, 0), 0);
		if (hold && refpose7) na += relative_angular(t, m, refpose7, hold, ang + na);
		for (int i = 0; st < 5 && i < ncray && i < 8; i++)
		{
			f4 q = quat_from_to(F3(0, 0, 1), F3(crays4[4 * i], crays4[4 * i + 1], crays4[4 * i + 2]));
			f3 off = F3(FEATURE[i].off[0], FEATURE[i].off[1], FEATURE[i].off[2]);
			f2 fl = { -100000.0f, 100000.0f };
			nl += ho_along_direction_deadzone(B, -1, F3(0, 0, 0), FEATURE[i].bone, off, qxdir(q), 0.01f, fl, lin + nl);
			nl += ho_along_direction_deadzone(B, -1, F3(0, 0, 0), FEATURE[i].bone, off, qydir(q), 0.01f, fl, lin + nl);
		}
		if (selectrb >= 0) nl += ho_position_nailed(B, -1, spoint, selectrb, rbpoint, lin + nl);
		if (st < steps_ - 1)
		{
			int c0 = nl;
			nl += cloud_constraints(m, points, n, 1, F3(0, 0, 0), lin + nl);
			for (int i = c0; i < nl; i++)
			{
				const float k = t->par.microforce * (1.0f * (steps_ - st) / (float)steps_) * ((lin[i].rb1 == 0) ? 0.1f * (st < steps_ - 2) : 1.0f);
				lin[i].forcelimit.x *= k; lin[i].forcelimit.y *= k;
			}
		}
		ho_fit_pointcloud(t, m, NULL, 0, lin, nl, ang, na, 1.0f);
	}
	free(lin);
}
