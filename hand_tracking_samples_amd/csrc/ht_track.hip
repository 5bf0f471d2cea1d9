// ht_track.hip -- tracker-level kernels: state plumbing, the full-reset path (PoseFromScratch + UnibodyFit), accept/reject of the
// CNN-driven pose and the user-space pose output.
//
// Reference computations:
//   PoseFromScratch / FixPositions / Reset    include/handtrack.h:480-506, include/physmodel.h:404-408, 221-226
//   UnibodyFit                                 include/handtrack.h:451-470
//   update_cnn_model accept logic              include/handtrack.h:704-726
//   PhysModel::GetPoseUser / SetPose           include/physmodel.h:433-435, third_party/physics.h:142
#include "ht_device.hpp"
#include "ht_launch.hpp"
#include "ht_quad.hpp"

__device__ __forceinline__ v3 G3(const float *p) { return V3(p[0], p[1], p[2]); }
__device__ __forceinline__ v4 G4(const float *p) { return V4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ m3 GM(const float *p) { m3 m; m.x = V3(p[0], p[1], p[2]); m.y = V3(p[3], p[4], p[5]); m.z = V3(p[6], p[7], p[8]); return m; }

// ---- state plumbing ---------------------------------------------------------------------------------------------
// mode 0: SetPose from poses[B][nb][7] (momenta untouched); 1: SetPose + zero momenta; 2: copy pose from another state array; 3: full 13-float state in
__global__ void k_set_pose(float *__restrict__ state, const float *__restrict__ src, int nb, int n, int mode)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * nb) return;
	float *s = state + (size_t)i * HT_STATE_STRIDE;
	if (mode == 2) { const float *o = src + (size_t)i * HT_STATE_STRIDE; for (int k = 0; k < 7; k++) s[k] = o[k]; return; }
	if (mode == 3) { const float *o = src + (size_t)i * HT_STATE; for (int k = 0; k < 13; k++) s[k] = o[k]; return; }
	const float *o = src + (size_t)i * HT_POSE;
	for (int k = 0; k < 7; k++) s[k] = o[k];
	if (mode == 1) for (int k = 7; k < 13; k++) s[k] = 0.0f;
}
__global__ void k_get_state(const float *__restrict__ state, float *__restrict__ dst, int nb, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * nb) return;
	const float *s = state + (size_t)i * HT_STATE_STRIDE;
	for (int k = 0; k < 13; k++) dst[(size_t)i * HT_STATE + k] = s[k];
}
__global__ void k_clear_flags(float *prev_err, int *initializing, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { prev_err[i] = 0.0f; initializing[i] = 0; }
}
// (the full-reset decision of handtrack.h:706 and the accept step of :713-731 ride on k_fit_error's last thread: ht_fit_after, csrc/ht_cloud.hip)

// ---- PoseFromScratch: one wave per flagged frame -----------------------------------------------------------------
__global__ __launch_bounds__(64) void k_scratch(ht_model_dev M, float *__restrict__ state, const float4 *__restrict__ pts, const int *__restrict__ npts,
                                               const float *__restrict__ analysis, const float *__restrict__ cams, const int *__restrict__ flags)
{
	__shared__ float pos[HT_MAXNB][3], q[HT_MAXNB][4];
	__shared__ float pc[3];
	const int b = blockIdx.x, lane = threadIdx.x;
	if (flags && !flags[b]) return;
	const float *an = analysis + (size_t)b * HT_ANALYSIS;
	const float *cam = cams + (size_t)b * HT_CAM;
	{
		// palm ray from the first three landmark rays, inverse-distance weighted centroid of the cloud (handtrack.h:483-490).  The weights are
		// independent per point and are formed by all lanes (256 points per pass, through LDS); the sums keep the reference's order on lane 0.
		__shared__ float4 wp[256];
		const v4 cs = (G4(an + HT_AN_CRAYS) + G4(an + HT_AN_CRAYS + 4)) + G4(an + HT_AN_CRAYS + 8);
		const v3 palmray = normalize(xyz(cs));
		v3 pcom = V3(0, 0, 0); float wsum = 0.00000000001f;
		const int n = npts[b];
		for (int base = 0; base < n; base += 256)
		{
			const int m = min(256, n - base);
			__syncthreads();
			for (int i = lane; i < m; i += 64)
			{
				const float4 pv = pts[(size_t)b * M.pts_cap + base + i];
				const v3 p = V3(pv.x, pv.y, pv.z);
				const v3 c = cross(p, palmray);
				const float w = 1.0f / (0.000001f + dot(c, c));
				const v3 pw = p * w;
				wp[i] = make_float4(pw.x, pw.y, pw.z, w);
			}
			__syncthreads();
			if (lane == 0)      // the reference's order of additions; eight terms are read ahead of the (dependent) sums
			{
				int i = 0;
				for (; i + 8 <= m; i += 8)
				{
					float4 e[8];
#pragma unroll
					for (int k = 0; k < 8; k++) e[k] = wp[i + k];
#pragma unroll
					for (int k = 0; k < 8; k++) { pcom = pcom + V3(e[k].x, e[k].y, e[k].z); wsum += e[k].w; }
				}
				for (; i < m; i++) { const float4 e = wp[i]; pcom = pcom + V3(e.x, e.y, e.z); wsum += e.w; }
			}
		}
		if (lane == 0) { pcom = pcom / wsum; pc[0] = pcom.x; pc[1] = pcom.y; pc[2] = pcom.z; }
	}
	if (lane < M.nb)
	{
		const float *bc = M.bodyc + lane * HT_BC;     // Reset(rb) physmodel.h:221-226
		for (int i = 0; i < 3; i++) pos[lane][i] = bc[HT_BC_POS0 + i];
		for (int i = 0; i < 4; i++) q[lane][i] = bc[HT_BC_Q0 + i];
	}
	__syncthreads();
	const v4 camq = V4(cam[8], cam[9], cam[10], cam[11]);
	const v4 palmq = G4(an + HT_AN_PALMQ);
	xf p1 = XF(V3(pc[0], pc[1], pc[2]), qmul(camq, palmq));
	xf dp = mul(p1, inverse(XF(G3(pos[1]), G4(q[1]))));
	__syncthreads();
	if (lane < M.nb)
	{
		xf np = mul(dp, XF(G3(pos[lane]), G4(q[lane])));
		pos[lane][0] = np.p.x; pos[lane][1] = np.p.y; pos[lane][2] = np.p.z; q[lane][0] = np.q.x; q[lane][1] = np.q.y; q[lane][2] = np.q.z; q[lane][3] = np.q.w;
	}
	__syncthreads();
	if (lane >= 1 && lane <= 4 && M.nb >= 17)     // curl the four fingers by the decoded clench angles (handtrack.h:498-504)
	{
		const int finger = lane;
		const float a = an[HT_AN_CLENCH + finger];
		const v4 jf = G4(M.jointc + (1 + finger * 3) * HT_JC + HT_JC_FRAME);
		const float ang[3] = { a / 2.0f, a, a * 1.25f };
		for (int k = 0; k < 3; k++)
		{
			const int bb = 2 + k + finger * 3;
			v4 o = qmul(jf, qmul(G4(q[bb]), quat_axis_angle(V3(1, 0, 0), ang[k])));
			q[bb][0] = o.x; q[bb][1] = o.y; q[bb][2] = o.z; q[bb][3] = o.w;
		}
	}
	__syncthreads();
	// FixPositions: ordered top-down (physmodel.h:404-408).  The joints' constants come in by one lane per joint first (a lane walking them alone waited
	// two dependent memory round trips per joint), then lane 0 walks the chain in order out of LDS.
	__shared__ float fj[HT_MAXNJ][12];      // rb0, rb1, p0 (3), p1 (3), then the two bodies' centres of mass are folded below
	__shared__ float fc[HT_MAXNJ][6];
	if (lane < M.nj)
	{
		const float *jc = M.jointc + lane * HT_JC;
		const int r0 = (int)jc[HT_JC_RB0], r1 = (int)jc[HT_JC_RB1];
		fj[lane][0] = (float)r0; fj[lane][1] = (float)r1;
		for (int i = 0; i < 3; i++) { fj[lane][2 + i] = jc[HT_JC_P0 + i]; fj[lane][5 + i] = jc[HT_JC_P1 + i]; }
		for (int i = 0; i < 3; i++) { fc[lane][i] = M.bodyc[r0 * HT_BC + HT_BC_COM + i]; fc[lane][3 + i] = M.bodyc[r1 * HT_BC + HT_BC_COM + i]; }
	}
	__syncthreads();
	if (lane == 0)
	{
		for (int j = 0; j < M.nj; j++)
		{
			const int r0 = (int)fj[j][0], r1 = (int)fj[j][1];
			xf u0 = XF(apply(XF(G3(pos[r0]), G4(q[r0])), -G3(fc[j])), G4(q[r0]));
			xf u1 = XF(apply(XF(G3(pos[r1]), G4(q[r1])), -G3(fc[j] + 3)), G4(q[r1]));
			v3 np = G3(pos[r1]) + (apply(u0, G3(fj[j] + 2)) - apply(u1, G3(fj[j] + 5)));
			pos[r1][0] = np.x; pos[r1][1] = np.y; pos[r1][2] = np.z;
		}
	}
	__syncthreads();
	if (lane < M.nb)
	{
		float *s = state + ((size_t)b * M.nb + lane) * HT_STATE_STRIDE;
		for (int i = 0; i < 3; i++) s[i] = pos[lane][i];
		for (int i = 0; i < 4; i++) s[3 + i] = q[lane][i];
		for (int i = 7; i < 13; i++) s[i] = 0.0f;
	}
}

// ---- UnibodyFit: one wave per flagged frame; the rows all act on one proxy body, so the Gauss-Seidel chain is sequential -----------
__global__ __launch_bounds__(64) void k_unibody(ht_model_dev M, ht_physics_dev ph, float *__restrict__ state, const float *__restrict__ rows, const int *__restrict__ nrows,
                                                const int *__restrict__ flags, float *__restrict__ scratch, int scratch_stride, int batch)
{
	__shared__ float pos[HT_MAXNB][3], q[HT_MAXNB][4];
	__shared__ float res[8];
	const int b = blockIdx.x, lane = threadIdx.x;
	if (flags && !flags[b]) return;
	const int nb = M.nb;
	float *st = state + (size_t)b * nb * HT_STATE_STRIDE;
	if (lane < nb) { for (int i = 0; i < 3; i++) pos[lane][i] = st[lane * HT_STATE_STRIDE + i]; for (int i = 0; i < 4; i++) q[lane][i] = st[lane * HT_STATE_STRIDE + 3 + i]; }
	__syncthreads();
	// SanityCheck before the solve (handtrack.h:463) is a no-op unless the pose already holds NaNs; those bodies are reset
	if (lane < nb)
	{
		bool bad = false;
		for (int i = 0; i < 13; i++) bad = bad || isnan(st[lane * HT_STATE_STRIDE + i]);
		if (bad) { for (int i = 0; i < 3; i++) pos[lane][i] = M.bodyc[lane * HT_BC + HT_BC_POS0 + i]; for (int i = 0; i < 4; i++) q[lane][i] = M.bodyc[lane * HT_BC + HT_BC_Q0 + i]; }
	}
	__syncthreads();
	const int n = nrows[b];
	const float dt = ph.deltaT;
	const v3 ubpos = G3(pos[1]) + V3(M.ub_com[0], M.ub_com[1], M.ub_com[2]);        // RigidBody ctor: position += com (physics.h:157)
	const v4 ubq = G4(q[1]);
	const xf ubi = inverse(XF(ubpos, ubq));
	const float minv = M.ub_massinv;
	const m3 tinv = GM(M.ub_tinv);
	const m3 Iinv = world_inertia(ubq, tinv, minv);
	// rows of this solve (every 4th point): re-expressed on the proxy body and pre-computed into the frame's record stream (handtrack.h:457-462).
	// Up to UB_LDS_ROWS rows (3584 points) the records stay in LDS (77 KB: only the few frames that take the full-reset
	// path run this kernel, so occupancy is no concern, and a single quad walking its chain alone on a CU would wait a whole L2 round trip for what
	// k_solve's sixteen quads overlap); a larger cloud uses the frame's slot of the solver scratch in HBM, sums behind all frames' records as in k_solve.
	constexpr int UB_LDS_ROWS = 896;      // 66 KB: with four 20 KB solver blocks on a CU (1024 frames) this block still finds room at once
	__shared__ __attribute__((aligned(16))) float urow[(UB_LDS_ROWS + QUAD_CHAIN_SLACK) * CREC];
	__shared__ float usum[UB_LDS_ROWS + QUAD_CHAIN_SLACK];      // impulse sums of the rows
	__shared__ unsigned short uidx[UB_LDS_ROWS + QUAD_CHAIN_SLACK];      // the chain as quad_chain_run walks it: here simply every record in order
	const int nr = n < scratch_stride - QUAD_CHAIN_SLACK ? n : scratch_stride - QUAD_CHAIN_SLACK;
	const bool in_lds = nr <= UB_LDS_ROWS;
	float *const grec = scratch + (size_t)b * scratch_stride * CREC;
	float *const gsum = scratch + (size_t)batch * scratch_stride * CREC + (size_t)b * scratch_stride;
	unsigned *const gidx = reinterpret_cast<unsigned *>(scratch + (size_t)batch * scratch_stride * (CREC + 1)) + (size_t)b * scratch_stride;
	if (in_lds) { for (int i = lane; i < nr + QUAD_CHAIN_SLACK; i += 64) { usum[i] = 0.0f; uidx[i] = (unsigned short)i; } }
	else for (int i = lane; i < nr + QUAD_CHAIN_SLACK; i += 64) { gsum[i] = 0.0f; gidx[i] = (unsigned)i; }
	for (int i = lane; i < nr; i += 64)
	{
		const float *r = rows + ((size_t)b * M.pts_cap + i) * HT_ROW;
		const int rb1 = (int)r[1];
		const v3 p1 = apply(ubi, apply(XF(G3(pos[rb1]), G4(q[rb1])), G3(r + 5)));
		const v3 nrm = G3(r + 8);
		const v3 r1 = qrot(ubq, p1);
		const float impulsed = minv + dot(cross(mul(Iinv, cross(r1, nrm)), r1), nrm);
		const float ts = r[11] / dt;
		quad_write_record((in_lds ? urow : grec) + (size_t)i * CREC, r1, nrm, Iinv, minv, ts, fmin_std(ts, r[12]), impulsed, r[13] * dt, r[14] * dt);
	}
	__threadfence_block();
	__syncthreads();
	if (lane < 4)       // the proxy body in quad layout (ht_quad.hpp): lane c < 3 owns component c, lane 3 carries the row's target speed
	{
		const int c = lane;
		// rbinitvelocity on a body at rest: 0 * damping + 0
		quad_body qb = { (0.0f * M.ub_dampleft) + 0.0f, (0.0f * M.ub_dampleft) + 0.0f };
		v3 pn = ubpos; v4 qn = ubq;
		const int total = ph.iterations + ph.iterations_post;
		for (int sweep = 0; sweep < total; sweep++)
		{
			const int tsoff = sweep >= ph.iterations ? 1 : 0;        // RemoveBias: lane 3 switches to the ts_post slot
			if (nr > 0) { if (in_lds) quad_chain_run(qb, urow, uidx, usum, nr, c, tsoff); else quad_chain_run(qb, grec, gidx, gsum, nr, c, tsoff); }
			if (sweep + 1 == ph.iterations)
			{
				const v3 lin = V3(dpp<QP_BC0>(qb.l), dpp<QP_BC1>(qb.l), dpp<QP_BC2>(qb.l)), ang = V3(dpp<QP_BC0>(qb.av), dpp<QP_BC1>(qb.av), dpp<QP_BC2>(qb.av));
				pn = ubpos + (lin * minv) * dt;
				const m3 tm = tinv * minv;
				auto diffq = [&](v4 o) -> v4 { v4 sn = normalize(o); m3 Mx = qmat(sn); m3 Ii = mul(Mx, mul(tm, transpose(Mx))); v3 hs = mul(Ii, ang) * 0.5f; return qmul(V4(hs.x, hs.y, hs.z, 0), sn); };
				v4 d1 = diffq(ubq), d2 = diffq(ubq + d1 * (dt / 2)), d3 = diffq(ubq + d2 * (dt / 2)), d4 = diffq(ubq + d3 * dt);
				v4 o = normalize((((ubq + d1 * (dt / 6)) + d2 * (dt / 3)) + d3 * (dt / 3)) + d4 * (dt / 6));
				if (o.x < FLT_EPSILON / 4.0f && o.x > -FLT_EPSILON / 4.0f) o.x = 0.0f;
				if (o.y < FLT_EPSILON / 4.0f && o.y > -FLT_EPSILON / 4.0f) o.y = 0.0f;
				if (o.z < FLT_EPSILON / 4.0f && o.z > -FLT_EPSILON / 4.0f) o.z = 0.0f;
				qn = o;
			}
		}
		if (lane == 0) {
		res[0] = pn.x; res[1] = pn.y; res[2] = pn.z; res[3] = qn.x; res[4] = qn.y; res[5] = qn.z; res[6] = qn.w; }
	}
	__syncthreads();
	const xf dp = mul(XF(V3(res[0], res[1], res[2]), V4(res[3], res[4], res[5], res[6])), inverse(XF(G3(pos[1]), G4(q[1]))));
	if (lane < nb)
	{
		xf np = mul(dp, XF(G3(pos[lane]), G4(q[lane])));
		float *s = st + lane * HT_STATE_STRIDE;
		bool bad = isnan(np.p.x) || isnan(np.p.y) || isnan(np.p.z) || isnan(np.q.x) || isnan(np.q.y) || isnan(np.q.z) || isnan(np.q.w);
		for (int i = 7; i < 13; i++) bad = bad || isnan(s[i]);
		if (bad) { const float *bc = M.bodyc + lane * HT_BC; np = XF(G3(bc + HT_BC_POS0), G4(bc + HT_BC_Q0)); for (int i = 7; i < 13; i++) s[i] = 0.0f; }
		s[0] = np.p.x; s[1] = np.p.y; s[2] = np.p.z; s[3] = np.q.x; s[4] = np.q.y; s[5] = np.q.z; s[6] = np.q.w;
	}
}

// ---- the user poses when no main pass follows (with one, its last k_solve writes them) -----------------------------------------------
// GetPoseUser (physmodel.h:434) + the "initializing = 50" rule of handtrack.h:781-782
// raw != 0: GetPose (physmodel.h:433) of the given model, no rule
__global__ void k_output(ht_model_dev M, const float *__restrict__ hand, const int *__restrict__ npts, int *__restrict__ initializing, int min_point_num, float *__restrict__ poses, int n, int raw)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * M.nb) return;
	const int b = i / M.nb, rb = i % M.nb;
	const float *s = hand + (size_t)i * HT_STATE_STRIDE;
	if (raw) { for (int k = 0; k < 7; k++) poses[(size_t)i * HT_POSE + k] = s[k]; return; }
	v3 pu = apply(XF(G3(s), G4(s + 3)), -G3(M.bodyc + rb * HT_BC + HT_BC_COM));
	float *o = poses + (size_t)i * HT_POSE;
	o[0] = pu.x; o[1] = pu.y; o[2] = pu.z; o[3] = s[3]; o[4] = s[4]; o[5] = s[5]; o[6] = s[6];
	if (rb == 0 && npts[b] < min_point_num) initializing[b] = 50;
}

// ---- launchers ----------------------------------------------------------------------------------------------------
void ht_launch_set_pose(float *state, const float *src, int nb, int n, int mode, hipStream_t s) { hipLaunchKernelGGL(k_set_pose, dim3((n * nb + 255) / 256), dim3(256), 0, s, state, src, nb, n, mode); }
void ht_launch_get_state(const float *state, float *dst, int nb, int n, hipStream_t s) { hipLaunchKernelGGL(k_get_state, dim3((n * nb + 255) / 256), dim3(256), 0, s, state, dst, nb, n); }
void ht_launch_clear_flags(float *prev_err, int *initializing, int n, hipStream_t s) { hipLaunchKernelGGL(k_clear_flags, dim3((n + 255) / 256), dim3(256), 0, s, prev_err, initializing, n); }
void ht_launch_scratch(const ht_model_dev &M, float *state, const float4 *pts, const int *npts, const float *analysis, const float *cams, const int *flags, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_scratch, dim3(B), dim3(64), 0, s, M, state, pts, npts, analysis, cams, flags);
}
// PhysModel::scale, the pose part (physmodel.h:312-313): rb.position = rb0.position + (rb.position - rb0.position) * s
__global__ void k_scale_state(float *__restrict__ state, int nb, int n, float s)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * nb) return;
	const int f = i / nb, b = i % nb;
	const float *p0 = state + (size_t)f * nb * HT_STATE_STRIDE;
	float *p = state + ((size_t)f * nb + b) * HT_STATE_STRIDE;
	if (b == 0) return;      // p0 + (p0 - p0) * s = p0
	for (int k = 0; k < 3; k++) p[k] = p0[k] + (p[k] - p0[k]) * s;
}
void ht_launch_scale_state(float *state, int nb, int n, float s, hipStream_t st)
{
	hipLaunchKernelGGL(k_scale_state, dim3((n * nb + 255) / 256), dim3(256), 0, st, state, nb, n, s);
}
void ht_launch_unibody(const ht_model_dev &M, const ht_physics_dev &ph, float *state, const float *rows, const int *nrows, const int *flags, float *scratch, int scratch_stride, int batch, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_unibody, dim3(B), dim3(64), 0, s, M, ph, state, rows, nrows, flags, scratch, scratch_stride, batch);
}
void ht_launch_output(const ht_model_dev &M, const float *hand, const int *npts, int *initializing, int min_point_num, float *poses, int n, hipStream_t s, int raw)
{
	hipLaunchKernelGGL(k_output, dim3((n * M.nb + 255) / 256), dim3(256), 0, s, M, hand, npts, initializing, min_point_num, poses, n, raw);
}
