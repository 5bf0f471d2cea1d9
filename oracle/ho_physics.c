/*
 * ho_physics.c -- CPU restatement of the rigid-body constraint solver the reference runs per fit step
 * (third_party/physics.h) and of the constraint factories the hand tracker uses.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  Each function cites the reference lines it follows and keeps the
 * reference's float evaluation order (including the places where the reference silently computes in double
 * because an unqualified sin()/acos()/cos() resolves to the C double overload).
 */
#include <stdlib.h>
#include <string.h>
#include "ht_oracle.h"

static inline pose_t body_pose(const ho_body *b) { return POSE(b->position, b->orientation); }
static inline f3 body_spin(const ho_body *b) { return m33_mulv(b->Iinv, b->angmom); }            /* physics.h:126 */

/* ---- LimitLinear / LimitAngular constructors (physics.h:248-249, 283-287) ---- */
static ho_linear mk_linear(int rb0, int rb1, f3 p0, f3 p1, f3 normal, float targetdist, float tsnb, float fmin, float fmax)
{
	ho_linear l;
	memset(&l, 0, sizeof l);
	l.rb0 = rb0; l.rb1 = rb1; l.position0 = p0; l.position1 = p1; l.normal = normal; l.targetdist = targetdist; l.targetspeednobias = tsnb;
	l.forcelimit.x = ho_minf(fmin, fmax); l.forcelimit.y = ho_maxf(fmin, fmax);
	l.friction_master = 0; l.impulsesum = 0;
	return l;
}
static ho_angular mk_angular(int rb0, int rb1, f3 axis, float targetspin, float mintorque, float maxtorque)
{
	ho_angular a; a.rb0 = rb0; a.rb1 = rb1; a.axis = axis; a.torque = 0; a.targetspin = targetspin; a.mintorque = mintorque; a.maxtorque = maxtorque; return a;
}
static inline f3 anchor_world(ho_body *const *B, int rb, f3 p) { return rb >= 0 ? pose_apply(body_pose(B[rb]), p) : p; }

/* physics.h:328-331 */
ho_linear ho_along_direction(ho_body *const *B, int rb0, f3 p0, int rb1, f3 p1, f3 axisw, float minforce, float maxforce)
{
	return mk_linear(rb0, rb1, p0, p1, axisw, dot3(sub3(anchor_world(B, rb1, p1), anchor_world(B, rb0, p0)), axisw), 0.0f, minforce, maxforce);
}
/* physics.h:332-340 */
int ho_along_direction_deadzone(ho_body *const *B, int rb0, f3 p0, int rb1, f3 p1, f3 axisw, float radius, f2 forcelimit, ho_linear *out)
{
	out[0] = mk_linear(rb0, rb1, p0, p1, axisw, dot3(sub3(anchor_world(B, rb1, p1), anchor_world(B, rb0, p0)), axisw) + radius, 0.0f, 0, forcelimit.y);
	out[1] = mk_linear(rb0, rb1, p0, p1, axisw, dot3(sub3(anchor_world(B, rb1, p1), anchor_world(B, rb0, p0)), axisw) - radius, 0.0f, forcelimit.x, 0);
	return 2;
}
/* physics.h:342-346 */
int ho_position_nailed(ho_body *const *B, int rb0, f3 p0, int rb1, f3 p1, ho_linear *out)
{
	f3 d = sub3(anchor_world(B, rb1, p1), anchor_world(B, rb0, p0));
	out[0] = mk_linear(rb0, rb1, p0, p1, F3(1, 0, 0), d.x, 0.0f, -FLT_MAX, FLT_MAX);
	out[1] = mk_linear(rb0, rb1, p0, p1, F3(0, 1, 0), d.y, 0.0f, -FLT_MAX, FLT_MAX);
	out[2] = mk_linear(rb0, rb1, p0, p1, F3(0, 0, 1), d.z, 0.0f, -FLT_MAX, FLT_MAX);
	return 3;
}
/* geometric.h:218-232 maxdir: first maximum of dot(v,dir) */
int ho_maxdir(const f3 *p, int count, f3 dir)
{
	int best = 0;
	for (int i = 1; i < count; i++) if (dot3(p[best], dir) < dot3(p[i], dir)) best = i;
	return best;
}
/* physics.h:347-350 */
ho_linear ho_under_plane(ho_body *const *B, int rb, f4 plane, float maxforce)
{
	const ho_body *b = B[rb];
	f3 dirl = qrot(qconj(b->orientation), xyz(plane));
	f3 sv = b->shape.verts[ho_maxdir(b->shape.verts, b->shape.nverts, dirl)];
	return ho_along_direction(B, -1, scale3(xyz(plane), -plane.w), rb, sv, neg3(xyz(plane)), 0, maxforce);
}

/* physics.h:351-393.  sin() there is the C double overload: the sums are formed in double and rounded once. */
static int angular_range_w(const ho_physics *ph, int rb0, f4 jb0, int rb1, f4 jf1, f3 lmin, f3 lmax, ho_angular *out)
{
	int n = 0;
	float dt = ph->deltaT;
	f3 jmin = div3(scale3(lmin, 3.14f), 180.0f);
	f3 jmax = div3(scale3(lmax, 3.14f), 180.0f);
	if (jmin.x == 0 && jmax.x == 0 && jmin.z < jmax.z)
	{
		f4 cb = normalize4(F4(0, -1, 0, 1));
		return angular_range_w(ph, rb0, qmul(jb0, cb), rb1, qmul(jf1, cb), F3(lmin.z, lmin.y, 0), F3(lmax.z, lmax.y, 0), out);
	}
	f4 r = qmul(qconj(jb0), jf1);
	f4 s = quat_from_to(F3(0, 0, 1.0f), qzdir(r));
	f4 t = qmul(qconj(s), r);
	if (jmax.x == jmin.x)
		out[n++] = mk_angular(rb0, rb1, qxdir(jf1), (float)(2 * ((double)(-s.x) + sin((double)(jmin.x / 2.0f))) / (double)dt), -FLT_MAX, FLT_MAX);
	else if (jmax.x - jmin.x < 360.0f * 3.14f / 180.0f)
	{
		out[n++] = mk_angular(rb0, rb1, qxdir(jf1), (float)(2 * ((double)(-s.x) + sin((double)(jmin.x / 2.0f))) / (double)dt), 0, FLT_MAX);
		out[n++] = mk_angular(rb0, rb1, neg3(qxdir(jf1)), (float)(2 * ((double)(s.x) - sin((double)(jmax.x / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	if (jmax.y == jmin.y)
		out[n++] = mk_angular(rb0, rb1, qydir(jf1), ph->biasfactorjoint * 2 * (-s.y + jmin.y) / dt, -FLT_MAX, FLT_MAX);
	else
	{
		out[n++] = mk_angular(rb0, rb1, qydir(jf1), (float)(2 * ((double)(-s.y) + sin((double)(jmin.y / 2.0f))) / (double)dt), 0, FLT_MAX);
		out[n++] = mk_angular(rb0, rb1, neg3(qydir(jf1)), (float)(2 * ((double)(s.y) - sin((double)(jmax.y / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	if (jmin.z == jmax.z)
		out[n++] = mk_angular(rb0, rb1, qzdir(jf1), ph->biasfactorjoint * 2 * -t.z / dt, -FLT_MAX, FLT_MAX);
	else
	{
		out[n++] = mk_angular(rb0, rb1, qzdir(jf1), (float)(2 * ((double)(-t.z) + sin((double)(jmin.z / 2.0f))) / (double)dt), 0, FLT_MAX);
		out[n++] = mk_angular(rb0, rb1, neg3(qzdir(jf1)), (float)(2 * ((double)(t.z) - sin((double)(jmax.z / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	return n;
}
/* physics.h:395-399 */
int ho_angular_range(const ho_physics *ph, ho_body *const *B, int rb0, int rb1, f4 jointframe, f3 lmin, f3 lmax, ho_angular *out)
{
	return angular_range_w(ph, rb0, rb0 >= 0 ? qmul(B[rb0]->orientation, jointframe) : jointframe, rb1, rb1 >= 0 ? B[rb1]->orientation : F4(0, 0, 0, 1), lmin, lmax, out);
}
/* physics.h:313-326 */
int ho_angular_drive(const ho_physics *ph, ho_body *const *B, int rb0, int rb1, f4 target, float maxtorque, ho_angular *out)
{
	f4 q0 = rb0 >= 0 ? B[rb0]->orientation : F4(0, 0, 0, 1);
	f4 q1 = rb1 >= 0 ? B[rb1]->orientation : F4(0, 0, 0, 1);
	f4 dq = qmul(q1, qconj(qmul(q0, target)));
	if (dq.w < 0) dq = neg4(dq);
	f3 axis = safenormalize3(xyz(dq));
	f3 binormal = ho_orth(axis);
	f3 normal = cross3(axis, binormal);
	out[0] = mk_angular(rb0, rb1, axis, -ph->biasfactorjoint * (ho_acosf(ho_clampf(dq.w, -1.0f, 1.0f)) * 2.0f) / ph->deltaT, -maxtorque, maxtorque);
	out[1] = mk_angular(rb0, rb1, binormal, 0, -maxtorque, maxtorque);
	out[2] = mk_angular(rb0, rb1, normal, 0, -maxtorque, maxtorque);
	return 3;
}
/* physics.h:402-414 */
ho_angular ho_cone_angle(const ho_physics *ph, ho_body *const *B, int rb0, f3 n0, int rb1, f3 n1, float limitangle_degrees)
{
	int equality = (limitangle_degrees == 0);
	f3 a0 = rb0 >= 0 ? qrot(B[rb0]->orientation, n0) : n0;
	f3 a1 = rb1 >= 0 ? qrot(B[rb1]->orientation, n1) : n1;
	f3 axis = safenormalize3(cross3(a1, a0));
	float rbangle = ho_acosf(ho_clampf(dot3(a0, a1), 0.0f, 1.0f));
	float dangle = rbangle - (limitangle_degrees) * 3.14f / 180.0f;
	float targetspin = ((equality) ? ph->biasfactorjoint : 1.0f) * dangle / ph->deltaT;
	return mk_angular(rb0, rb1, axis, targetspin, (limitangle_degrees > 0.0f) ? 0 : -FLT_MAX, FLT_MAX);
}

/* ---- contacts -> rows (physics.h:463-489) ---- */
int ho_constrain_contacts(const ho_physics *ph, ho_body *const *B, const ho_contact *C, int nc, ho_linear *out)
{
	int n = 0;
	for (int i = 0; i < nc; i++)
	{
		const ho_contact *c = &C[i];
		const ho_body *rb0 = c->rb0 >= 0 ? B[c->rb0] : NULL, *rb1 = c->rb1 >= 0 ? B[c->rb1] : NULL;
		f3 r0 = rb0 ? sub3(c->p0w, rb0->position) : F3(0, 0, 0);
		f3 v0 = rb0 ? add3(cross3(body_spin(rb0), r0), scale3(rb0->linmom, rb0->massinv)) : F3(0, 0, 0);
		f3 r1 = rb1 ? sub3(c->p1w, rb1->position) : F3(0, 0, 0);
		f3 v1 = rb1 ? add3(cross3(body_spin(rb1), r1), scale3(rb1->linmom, rb1->massinv)) : F3(0, 0, 0);
		f3 v = sub3(v0, v1);
		float minsep = ph->driftmax * 0.25f;
		float separation = c->separation;
		float bouncevel = ho_maxf(0.0f, (-dot3(c->normal, v) - length3(ph->gravity) * ph->falltime_to_ballistic) * ph->restitution);
		out[n++] = mk_linear(c->rb0, c->rb1, c->p0, c->p1, neg3(c->normal), ho_minf((separation - minsep) * ph->biasfactorpositive, separation), -bouncevel, 0, FLT_MAX);
		f4 q = quat_from_to(F3(0, 0, 1), neg3(c->normal));
		f3 tangent = qxdir(q), binormal = qydir(q);
		ho_linear fb = mk_linear(c->rb0, c->rb1, c->p0, c->p1, binormal, 0, 0, 0, 0); fb.friction_master = -1;
		ho_linear ft = mk_linear(c->rb0, c->rb1, c->p0, c->p1, tangent, 0, 0, 0, 0); ft.friction_master = -2;
		out[n++] = fb; out[n++] = ft;
	}
	return n;
}

/* ---- integrator pieces (physics.h:202-218, 500-541) ---- */
static f4 diffq(f4 orientation, m33 tensorinv, f3 angular)
{
	f4 sn = normalize4(orientation);
	m33 M = qmat(sn);
	m33 Iinv = m33_mul(M, m33_mul(tensorinv, m33_transpose(M)));
	f3 halfspin = scale3(m33_mulv(Iinv, angular), 0.5f);
	return qmul(F4(halfspin.x, halfspin.y, halfspin.z, 0), sn);
}
static f4 rkupdateq(f4 s, m33 tensorinv, f3 angular, float dt)
{
	f4 d1 = diffq(s, tensorinv, angular);
	f4 d2 = diffq(add4(s, scale4(d1, dt / 2)), tensorinv, angular);
	f4 d3 = diffq(add4(s, scale4(d2, dt / 2)), tensorinv, angular);
	f4 d4 = diffq(add4(s, scale4(d3, dt)), tensorinv, angular);
	return normalize4(add4(add4(add4(add4(s, scale4(d1, dt / 6)), scale4(d2, dt / 3)), scale4(d3, dt / 3)), scale4(d4, dt / 6)));
}
static void world_inertia(ho_body *rb) { m33 M = qmat(rb->orientation); rb->Iinv = m33_mul(M, m33_mul(m33_scale(rb->tensorinv_massless, rb->massinv), m33_transpose(M))); }
static void rbinitvelocity(const ho_physics *ph, ho_body *rb)
{
	float dampleftover = powf((1.0f - ho_maxf(rb->damping, ph->damping)), ph->deltaT);
	rb->linmom = scale3(rb->linmom, dampleftover);
	rb->angmom = scale3(rb->angmom, dampleftover);
	f3 force = scale3(scale3(ph->gravity, rb->mass), rb->gravscale);
	f3 torque = F3(0, 0, 0);
	rb->linmom = add3(rb->linmom, scale3(force, ph->deltaT));
	rb->angmom = add3(rb->angmom, scale3(torque, ph->deltaT));
	world_inertia(rb);
}
static void rbcalcnextpose(const ho_physics *ph, ho_body *rb)
{
	rb->position_next = add3(rb->position, scale3(scale3(rb->linmom, rb->massinv), ph->deltaT));
	f4 o = rkupdateq(rb->orientation, m33_scale(rb->tensorinv_massless, rb->massinv), rb->angmom, ph->deltaT);
	if (o.x < FLT_EPSILON / 4.0 && o.x > -FLT_EPSILON / 4.0) o.x = 0.0f;
	if (o.y < FLT_EPSILON / 4.0 && o.y > -FLT_EPSILON / 4.0) o.y = 0.0f;
	if (o.z < FLT_EPSILON / 4.0 && o.z > -FLT_EPSILON / 4.0) o.z = 0.0f;
	rb->orientation_next = o;
}
static void rbupdatepose(ho_body *rb) { rb->position = rb->position_next; rb->orientation = rb->orientation_next; world_inertia(rb); }

/* ---- Gauss-Seidel row updates (physics.h:251-265, 289-307) ---- */
static void angular_iter(const ho_physics *ph, ho_body *const *B, ho_angular *a)
{
	if (a->targetspin == -FLT_MAX) return;
	ho_body *rb0 = a->rb0 >= 0 ? B[a->rb0] : NULL, *rb1 = a->rb1 >= 0 ? B[a->rb1] : NULL;
	float currentspin = ((rb1) ? dot3(body_spin(rb1), a->axis) : 0.0f) - ((rb0) ? dot3(body_spin(rb0), a->axis) : 0.0f);
	float dspin = a->targetspin - currentspin;
	float spintotorque = 1.0f / (((rb0) ? dot3(a->axis, m33_mulv(rb0->Iinv, a->axis)) : 0.0f) + ((rb1) ? dot3(a->axis, m33_mulv(rb1->Iinv, a->axis)) : 0.0f));
	float dtorque = dspin * spintotorque;
	dtorque = ho_minf(dtorque, a->maxtorque * ph->deltaT - a->torque);
	dtorque = ho_maxf(dtorque, a->mintorque * ph->deltaT - a->torque);
	if (rb0) rb0->angmom = sub3(rb0->angmom, scale3(a->axis, dtorque));
	if (rb1) rb1->angmom = add3(rb1->angmom, scale3(a->axis, dtorque));
	a->torque += dtorque;
}
static void linear_iter(const ho_physics *ph, ho_body *const *B, ho_linear *rows, int i)
{
	ho_linear *l = &rows[i];
	ho_body *rb0 = l->rb0 >= 0 ? B[l->rb0] : NULL, *rb1 = l->rb1 >= 0 ? B[l->rb1] : NULL;
	if (l->friction_master)
		l->forcelimit.x = -(l->forcelimit.y = ho_maxf(((rb0) ? rb0->friction : 0), ((rb1) ? rb1->friction : 0)) * rows[i + l->friction_master].impulsesum / ph->deltaT);
	f3 r0 = (rb0) ? qrot(rb0->orientation, l->position0) : l->position0;
	f3 r1 = (rb1) ? qrot(rb1->orientation, l->position1) : l->position1;
	f3 v0 = (rb0) ? add3(cross3(body_spin(rb0), r0), scale3(rb0->linmom, rb0->massinv)) : F3(0, 0, 0);
	f3 v1 = (rb1) ? add3(cross3(body_spin(rb1), r1), scale3(rb1->linmom, rb1->massinv)) : F3(0, 0, 0);
	float vn = dot3(sub3(v1, v0), l->normal);
	float impulsen = -l->targetspeed - vn;
	float impulsed = ((rb0) ? rb0->massinv + dot3(cross3(m33_mulv(rb0->Iinv, cross3(r0, l->normal)), r0), l->normal) : 0)
	               + ((rb1) ? rb1->massinv + dot3(cross3(m33_mulv(rb1->Iinv, cross3(r1, l->normal)), r1), l->normal) : 0);
	float impulse = impulsen / impulsed;
	impulse = ho_minf(l->forcelimit.y * ph->deltaT - l->impulsesum, impulse);
	impulse = ho_maxf(l->forcelimit.x * ph->deltaT - l->impulsesum, impulse);
	if (rb0) { f3 imp = scale3(l->normal, -impulse); rb0->linmom = add3(rb0->linmom, imp); rb0->angmom = add3(rb0->angmom, cross3(r0, imp)); }
	if (rb1) { f3 imp = scale3(l->normal, impulse); rb1->linmom = add3(rb1->linmom, imp); rb1->angmom = add3(rb1->angmom, cross3(r1, imp)); }
	l->impulsesum += impulse;
}

static int hasnan3(f3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }
void ho_sanity_check(ho_model *m)   /* physmodel.h:221-229, 437-442 */
{
	for (int i = 0; i < m->nb; i++)
	{
		ho_body *rb = &m->bodies[i];
		if (hasnan3(rb->linmom) || hasnan3(rb->position) || hasnan3(rb->angmom) || hasnan3(xyz(rb->orientation)) || isnan(rb->orientation.w))
		{
			rb->position = rb->position_start; rb->orientation = rb->orientation_start; rb->linmom = rb->angmom = F3(0, 0, 0);
		}
	}
}

/* PhysicsUpdate, physics.h:543-587.  `bodies` plays the role of the rigidbodies pointer vector; rows index into it.
 * Collision constraints are appended to lin (capacity lincap) when a full model is given. */
void ho_physics_update(ho_tracker *t, ho_body **bodies, int nb, ho_model *model_for_collision, ho_linear *lin, int nlin, int lincap, ho_angular *ang, int nang)
{
	const ho_physics *ph = &t->phys;
	for (int i = 0; i < nb; i++) rbinitvelocity(ph, bodies[i]);
	t->last_ncontacts = 0;
	if (ph->use_collision && model_for_collision)
	{
		ho_contact contacts[256];
		int nc = ho_find_contacts(t, model_for_collision, contacts, 256);
		t->last_ncontacts = nc;
		if (nlin + 3 * nc > lincap) { nc = (lincap - nlin) / 3; }
		nlin += ho_constrain_contacts(ph, bodies, contacts, nc, lin + nlin);
	}
	for (int i = 0; i < nlin; i++) lin[i].targetspeed = lin[i].targetdist / ph->deltaT;
	for (int s = 0; s < ph->iterations; s++)
	{
		for (int i = 0; i < nlin; i++) linear_iter(ph, bodies, lin, i);
		for (int i = 0; i < nang; i++) angular_iter(ph, bodies, &ang[i]);
	}
	for (int i = 0; i < nb; i++) rbcalcnextpose(ph, bodies[i]);
	for (int i = 0; i < nlin; i++) lin[i].targetspeed = ho_minf(lin[i].targetspeed, lin[i].targetspeednobias);                       /* physics.h:288 */
	for (int i = 0; i < nang; i++) ang[i].targetspin = (ang[i].mintorque < 0) ? 0 : ho_minf(ang[i].targetspin, 0.0f);             /* physics.h:250 */
	for (int s = 0; s < ph->iterations_post; s++)
	{
		for (int i = 0; i < nlin; i++) linear_iter(ph, bodies, lin, i);
		for (int i = 0; i < nang; i++) angular_iter(ph, bodies, &ang[i]);
	}
	for (int i = 0; i < nb; i++) rbupdatepose(bodies[i]);
}
