// ht_cloud.hip -- point-cloud side of the pose solver on CDNA4: closest-feature search, cloud constraint rows,
// boundary ("chamber") planes and the fit-error metric.
//
// Reference computations:
//   closest / mostabove                include/physmodel.h:127-162
//   CloudConstraint(s), ConvexHitCheck include/physmodel.h:164-181, third_party/geometric.h:275-302, physics.h:328-331
//   containing_plane / cloud_chamber   include/physmodel.h:183-193, 486-496, physics.h:347-350
//   FitError                           include/handtrack.h:371-399
//
// Mapping: one lane per point (64 points per wave).  The per-body loop and the per-plane loops are wave-uniform, so plane
// coefficients arrive through scalar loads and every active lane does the same 6 flops per plane; the sphere cull of
// physmodel.h:153 is a per-lane predicate.  Body poses are expanded once per block into an LDS table (lane b <-> body b).
// All arithmetic keeps the reference's evaluation order (-ffp-contract=off), so rows are bit-identical to the CPU path.
#include <limits.h>
#include <mutex>
#include "ht_device.hpp"
#include "ht_launch.hpp"
#include <string.h>
#include "ht_quad.hpp"

extern __shared__ __attribute__((aligned(16))) float4 s_planes[];      // all face planes of the model, staged once per block (25 KB for the hand)
__device__ __forceinline__ void stage_planes(const ht_model_dev &M, int t, int nthreads)
{
	const int np = M.plane_off[M.nb];
	for (int i = t; i < np; i += nthreads) s_planes[i] = M.planes[i];
}
#define BT 32       // floats per body-table entry
// pos 0..2 | q 3..6 | radius 7 | rinner 8 | invp 9..11 (= qrot(qconj(q), -pos)) | RI columns 12..20 (qmat(qconj(q))) | RF columns 21..29 (qmat(q)) |
// 30, 31: first face plane and number of face planes of the body (as integers: a lane that picks its body at run time would otherwise fetch them from the
// kernel-argument segment with a global load, a memory round trip inside the pair loop)
__device__ __forceinline__ void body_table_build(const ht_model_dev &M, const float *__restrict__ st, float *tab, int lane)
{
	if (lane < M.nb)
	{
		const float *s = st + lane * HT_STATE_STRIDE;
		v3 pos = V3(s[0], s[1], s[2]); v4 q = V4(s[3], s[4], s[5], s[6]);
		v4 qc = qconj(q);
		v3 invp = qrot(qc, -pos);
		m3 ri = qmat(qc), rf = qmat(q);
		float *t = tab + lane * BT;
		t[0] = pos.x; t[1] = pos.y; t[2] = pos.z; t[3] = q.x; t[4] = q.y; t[5] = q.z; t[6] = q.w;
		t[7] = M.bodyc[lane * HT_BC + HT_BC_RADIUS]; t[8] = M.bodyc[lane * HT_BC + HT_BC_RINNER];
		t[9] = invp.x; t[10] = invp.y; t[11] = invp.z;
		t[12] = ri.x.x; t[13] = ri.x.y; t[14] = ri.x.z; t[15] = ri.y.x; t[16] = ri.y.y; t[17] = ri.y.z; t[18] = ri.z.x; t[19] = ri.z.y; t[20] = ri.z.z;
		t[21] = rf.x.x; t[22] = rf.x.y; t[23] = rf.x.z; t[24] = rf.y.x; t[25] = rf.y.y; t[26] = rf.y.z; t[27] = rf.z.x; t[28] = rf.z.y; t[29] = rf.z.z;
		t[30] = __int_as_float(M.plane_off[lane]); t[31] = __int_as_float(M.plane_off[lane + 1] - M.plane_off[lane]);
	}
}
__device__ __forceinline__ v3 tab_pos(const float *t) { return V3(t[0], t[1], t[2]); }
__device__ __forceinline__ int tab_plane0(const float *t) { return __float_as_int(t[30]); }
__device__ __forceinline__ int tab_nplanes(const float *t) { return __float_as_int(t[31]); }
__device__ __forceinline__ v3 tab_to_local(const float *t, v3 w)      // pose.inverse() * w  (geometric.h:119,122)
{
	v3 X = V3(t[12], t[13], t[14]), Y = V3(t[15], t[16], t[17]), Z = V3(t[18], t[19], t[20]);
	return V3(t[9], t[10], t[11]) + ((X * w.x + Y * w.y) + Z * w.z);
}
__device__ __forceinline__ v3 tab_rot(const float *t, v3 v)           // qrot(q, v)
{
	v3 X = V3(t[21], t[22], t[23]), Y = V3(t[24], t[25], t[26]), Z = V3(t[27], t[28], t[29]);
	return (X * v.x + Y * v.y) + Z * v.z;
}
__device__ __forceinline__ v3 tab_to_world(const float *t, v3 v) { return tab_pos(t) + tab_rot(t, v); }     // pose * v

// ---- closest(rigidbodies, v), physmodel.h:137-162, for a chunk of CH points at a time ---------------------------------------------------
// The reference walks the bodies twice per point: first the inner-sphere planes (cheap), then, for every body whose outer sphere is not
// farther than the best distance so far, the body's most-above face plane (92 dot products), keeping the minimum in body order.
// Here the expensive part, "which face plane of body b is point p most above" -- a pure function of (p, b) -- is evaluated for all
// candidate pairs of the chunk at once, FOUR LANES PER PAIR (each scans every fourth plane, the four partial first-maxima are merged in
// index order through DPP), and the order-dependent part (which bodies are considered, which minimum wins) is replayed per point
// afterwards exactly as the reference does it.  Candidates = bodies whose outer sphere passes the test against the best inner-sphere
// distance: a superset of what the reference considers (its bound only shrinks while it walks), so the replay, which applies the
// reference's own test with the running minimum, skips exactly the bodies the reference skips.
#define CH 128                     // points per chunk (lane-per-point phases use the first CH threads of the block)
#define PAIR_WIN 1024              // (point, body) pairs processed per window
struct closest_lds
{
	float4 v[CH];                                  // the chunk's points
	unsigned mask[CH];                             // candidate bodies of each point
	int poff[CH + 1];                              // exclusive prefix of the candidate counts
	unsigned short pair[PAIR_WIN];                 // window of the pair list: point | body << 8
	unsigned char face[CH][HT_MAXNB];              // most-above face of body b for point p
	int wsum[8];
};
// inner-sphere plane of body t for point v (physmodel.h:141-142) and the outer-sphere bound of the second loop (:153)
__device__ __forceinline__ v4 inner_plane(const float *t, v3 v) { const v3 n = safenormalize(v - tab_pos(t)); return V4(n, -dot(tab_pos(t), n) - t[8]); }
__device__ __forceinline__ float outer_bound(const float *t, v3 v) { return length(v - tab_pos(t)) - t[7]; }

// All threads of the block call this.  Threads t and t + CH carry the same point `v` (`exists`: the chunk has a point there; `active` = exists on the
// owner lane t < CH); on return rbmin / pmin / dmin are the reference's result on the owner lanes.
template <int NT>
__device__ __forceinline__ void closest_chunk(const ht_model_dev &M, const float *tab, closest_lds &L, bool active, bool exists, v3 v, int npmax, int &rbmin, v4 &pmin, float &dmin, long long *cyc = nullptr)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	long long tm = cyc ? clock64() : 0;
#define CC_MARK(k) if (cyc) { const long long tn = clock64(); cyc[k] += tn - tm; tm = tn; }
	// ---- A: inner-sphere walk, candidate mask.  Two lanes per point (t and t + CH carry the same point): the lower takes the bodies [0, h), the upper
	//      [h, nb); "the first strict minimum in body order" of the whole walk is the lower half's unless the upper half's is strictly smaller.  The
	//      candidate test needs the final bound, so the halves meet twice through LDS (the face table's bytes, free until phase B).
	static_assert(NT == 2 * CH, "phase A pairs lane t with lane t + CH");
	pmin = V4(0, 0, 0, FLT_MAX); dmin = dot_plane(pmin, v); rbmin = -1;
	unsigned mask = 0;
	{
		const bool up = t >= CH; const int tp = t & (CH - 1);
		const int hb = (M.nb + 1) >> 1, b0 = up ? hb : 0, b1 = up ? M.nb : hb;
		for (int b = b0; b < b1; b++)
		{
			const v4 p = inner_plane(tab + b * BT, v);
			const float d = dot_plane(p, v);
			if (d < dmin) { pmin = p; dmin = d; rbmin = b; }
		}
		float *X = reinterpret_cast<float *>(&L.face[0][0]) + 6 * tp;
		if (up) { X[0] = pmin.x; X[1] = pmin.y; X[2] = pmin.z; X[3] = pmin.w; X[4] = dmin; X[5] = __int_as_float(rbmin); }
		__syncthreads();
		if (!up)
		{
			const float du = X[4];
			if (du < dmin) { pmin = V4(X[0], X[1], X[2], X[3]); dmin = du; rbmin = __float_as_int(X[5]); }
			X[4] = dmin;
		}
		__syncthreads();
		const float bound = up ? X[4] : dmin;
		unsigned part = 0;
		if (exists) for (int b = b0; b < b1; b++) if (!(outer_bound(tab + b * BT, v) > bound)) part |= 1u << b;
		if (up) X[5] = __int_as_float((int)part);
		__syncthreads();
		if (!up)
		{
			mask = part | (unsigned)__float_as_int(X[5]);
			L.v[t] = make_float4(v.x, v.y, v.z, 0.0f);
			L.mask[t] = mask;
		}
	}      // (the exchange area becomes the face table again two barriers further down)
	CC_MARK(0)
	// exclusive prefix of the candidate counts over the chunk (CH = 2 waves)
	int cnt = __popc(mask), incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
	if (lane == 63 && wave < 8) L.wsum[wave] = incl;
	__syncthreads();
	int base = 0;
	for (int w = 0; w < wave; w++) base += L.wsum[w];
	if (t < CH) L.poff[t] = base + incl - cnt;
	if (t == CH - 1) L.poff[CH] = base + incl;
	__syncthreads();
	const int total = L.poff[CH];
	CC_MARK(1)
	if (cyc) cyc[5] += total;
	// ---- B: most-above face of every candidate pair, four lanes per pair ----
	for (int w0 = 0; w0 < total; w0 += PAIR_WIN)
	{
		if (t < CH)      // this point's pairs that fall into the window
		{
			int k = L.poff[t];
			unsigned m = mask;
			while (m) { const int b = __ffs(m) - 1; m &= m - 1; if (k >= w0 && k < w0 + PAIR_WIN) L.pair[k - w0] = (unsigned short)(t | (b << 8)); k++; }
		}
		__syncthreads();
		const int nwin = min(PAIR_WIN, total - w0);
		const int g = lane & 3;
		for (int q0 = wave * 16; q0 < nwin; q0 += (NT / 64) * 16)
		{
			const int q = q0 + (lane >> 2);
			const bool on = q < nwin;
			const unsigned e = on ? L.pair[q] : 0;
			const int pt = e & 255, b = e >> 8;
			const float4 pv = L.v[pt];
			const float *tb = tab + b * BT;
			const v3 vl = tab_to_local(tb, V3(pv.x, pv.y, pv.z));
			const float4 *pl = s_planes + tab_plane0(tb);
			const int np = on ? tab_nplanes(tb) : 0;
			const unsigned long long live = __ballot(on);
			// this lane's faces g, g + 4, ...: the first is taken as it is (std::max_element starts from it), the others replace it when strictly larger; four are
			// read ahead of their use.  When every pair of the wave sits on a body with the model's largest face count (one count for all bodies of the hand) the
			// faces up to the last round need no range test; a face index past a body's last reads the next body's planes or the slack behind the copy.
			// The dot product ((x vx + y vy) + z vz) + w with the x and y products as ONE packed multiply on the (even-aligned) first two registers of the plane
			// read -- left to itself the compiler packs y with z and copies both into an aligned pair first, two moves per face.  The face index is counted in
			// its wave-uniform part u (a scalar register; this lane's face is g + u), so that taking a face is a select on a scalar, not an add per face.
			typedef float f2 __attribute__((ext_vector_type(2)));
			const f2 vxy = { vl.x, vl.y };
			auto pdot = [&](const float4 f) -> float { const f2 q = f2{ f.x, f.y } * vxy; return ((q.x + q.y) + f.z * vl.z) + f.w; };
			float best = 0.0f; int bu = -1;      // bu: the uniform part of the best face's index (-1: none yet)
			if (g < np) { best = pdot(pl[g]); bu = 0; }
			const float4 *pg = pl + g;
			int u = 4;
			if (__all(!on || np == npmax))
				for (; u + 3 + 12 < npmax; u += 16)      // a bound all four lanes of a pair share (the loop below takes what it leaves)
				{
					const float4 f0 = pg[u], f1 = pg[u + 4], f2_ = pg[u + 8], f3 = pg[u + 12];
					const float d0 = pdot(f0), d1 = pdot(f1), d2 = pdot(f2_), d3 = pdot(f3);
					if (best < d0) { best = d0; bu = u; }
					if (best < d1) { best = d1; bu = u + 4; }
					if (best < d2) { best = d2; bu = u + 8; }
					if (best < d3) { best = d3; bu = u + 12; }
				}
			for (; u + g < npmax; u += 16)
			{
				const float4 f0 = pg[u], f1 = pg[u + 4], f2_ = pg[u + 8], f3 = pg[u + 12];
				const float d0 = pdot(f0), d1 = pdot(f1), d2 = pdot(f2_), d3 = pdot(f3);
				const int i = u + g;
				if (i < np && best < d0) { best = d0; bu = u; }
				if (i + 4 < np && best < d1) { best = d1; bu = u + 4; }
				if (i + 8 < np && best < d2) { best = d2; bu = u + 8; }
				if (i + 12 < np && best < d3) { best = d3; bu = u + 12; }
			}
			int bi = bu < 0 ? -1 : bu + g;
			(void)live;
			// merge the four partial results: larger value wins, equal values keep the lower index (the first maximum overall)
#pragma unroll
			for (int o = 1; o <= 2; o <<= 1)
			{
				const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
				const bool take = oi >= 0 && (bi < 0 || best < ob || (best == ob && oi < bi));
				if (take) { best = ob; bi = oi; }
			}
			if (on && g == 0) L.face[pt][b] = (unsigned char)bi;
		}
		__syncthreads();
	}
	CC_MARK(2)
	// ---- C: the reference's second walk (physmodel.h:151-160) with the faces found above ----
	if (t < CH && active)
	{
		for (unsigned m = mask; m; m &= m - 1)      // the candidate bodies in ascending order (a lane has two or three: the loop runs as long as the wave's longest list, not over all bodies)
		{
			const int b = __ffs(m) - 1;
			const float *tb = tab + b * BT;
			if (outer_bound(tb, v) > dmin) continue;
			const float4 f = s_planes[tab_plane0(tb) + L.face[t][b]];
			const v3 n = tab_rot(tb, V3(f.x, f.y, f.z));               // Pose::TransformPlane geometric.h:124
			const v4 p = V4(n, f.w - dot(tab_pos(tb), n));
			const float d = dot_plane(p, v);
			if (d < dmin) { pmin = p; dmin = d; rbmin = b; }
		}
	}
	__syncthreads();      // the chunk's LDS is free for the next one
	CC_MARK(3)
#undef CC_MARK
}

// ------------------------------------------------------------------------------------------------- k_cloud_rows
// mode 0: forcelimit (-1,1) (CloudConstraints as is)     1: FitPointCloud scaling (physmodel.h:347)
//      2: MultiStepSim scaling (handtrack.h:656,681)     3: UnibodyFit scaling (handtrack.h:461)     4: slowfit scaling (handtrack.h:815-816)
// One block per frame, CH points per pass: closest feature as above, then ConvexHitCheck (geometric.h:275-297) of the chosen body, one lane per
// point with the body's faces read per lane from the LDS copy (the clipping of a segment is sequential in the faces; points are independent).
#define CR_THREADS 256
// One frame's rows by one block (or by `ny` blocks that share its passes: block `by` takes every ny-th).  tab / L / wq / wI: the block's LDS (wq, wI only in
// record mode); the planes go to s_planes.  k_cloud_rows below and the full-reset kernel (k_reset) both run this.
__device__ __forceinline__ void cloud_rows_frame(const ht_model_dev &M, const float *__restrict__ state, const float4 *__restrict__ pts, const float *__restrict__ cams, int stride,
                                                 int use_cam_origin, int mode, float microforce, float weak_force, float cf_max_point, float cf_max_sum, float unibody_force,
                                                 float *__restrict__ rows, int *__restrict__ nrows, const cloud_records &rec, int dbg, int b, int n, int by, int ny,
                                                 float *tab, closest_lds &L, float (*wq)[4], float (*wI)[10])
{
	const int t = threadIdx.x;
	const int nsub = (n + stride - 1) / stride;
	if (t < 64) body_table_build(M, state + (size_t)b * M.nb * HT_STATE_STRIDE, tab, t);
	if (rec.scratch && t >= 64 && t < 64 + M.nb)      // rbinitvelocity's world inverse inertia (physics.h:517-518), the expression of k_solve's prologue
	{
		const int k = t - 64;
		const float *s = state + ((size_t)b * M.nb + k) * HT_STATE_STRIDE, *bc = M.bodyc + k * HT_BC;
		const v4 q = V4(s[3], s[4], s[5], s[6]);
		m3 T; T.x = V3(bc[HT_BC_TINV], bc[HT_BC_TINV + 1], bc[HT_BC_TINV + 2]); T.y = V3(bc[HT_BC_TINV + 3], bc[HT_BC_TINV + 4], bc[HT_BC_TINV + 5]); T.z = V3(bc[HT_BC_TINV + 6], bc[HT_BC_TINV + 7], bc[HT_BC_TINV + 8]);
		const m3 I = world_inertia(q, T, bc[HT_BC_MASSINV]);
		wq[k][0] = q.x; wq[k][1] = q.y; wq[k][2] = q.z; wq[k][3] = q.w;
		wI[k][0] = I.x.x; wI[k][1] = I.x.y; wI[k][2] = I.x.z; wI[k][3] = I.y.x; wI[k][4] = I.y.y; wI[k][5] = I.y.z; wI[k][6] = I.z.x; wI[k][7] = I.z.y; wI[k][8] = I.z.z; wI[k][9] = bc[HT_BC_MASSINV];
	}
	stage_planes(M, t, CR_THREADS);
	__syncthreads();
	const float *cam = cams + (size_t)b * HT_CAM;
	const v3 origin = use_cam_origin ? V3(cam[5], cam[6], cam[7]) : V3(0, 0, 0);
	int npmax = 0;
	for (int bb = 0; bb < M.nb; bb++) npmax = max(npmax, M.plane_off[bb + 1] - M.plane_off[bb]);
	for (int base = by * CH; base < nsub; base += ny * CH)
	{
		const int i = base + (t & (CH - 1));
		const bool exists = i < nsub, active = t < CH && exists;
		const float4 pv = exists ? pts[(size_t)b * M.pts_cap + i * stride] : make_float4(0, 0, 0, 0);
		const v3 v = V3(pv.x, pv.y, pv.z);
		int rb; v4 p; float dmin;
		closest_chunk<CR_THREADS>(M, tab, L, active, exists, v, npmax, rb, p, dmin);
		if (rb < 0) rb = 0;
		// ConvexHitCheck from the ray origin (geometric.h:275-297), only for the points that face away (physmodel.h:170).  Few points do, and the
		// reference's loop over the body's faces only ACTS on a face the segment does not lie behind (both ends outside: no hit; straddling: the outer
		// end is clipped to the face): the wanted points of the chunk are listed, a DPP row of sixteen lanes takes one point and looks at the next
		// sixteen faces at once, applies the FIRST of them that acts on the current segment and looks on from the face after it.
		// Same faces, same order, same arithmetic on the same segment as the sequential loop -- which ran all 92 faces on two whole waves whenever any
		// of their points faced away (27 % of the kernel).
		const bool want = t < CH && active && dot(v - origin, xyz(p)) > 0 && !HT_DBG(dbg, 0x100000);
		const unsigned long long wm = __ballot(want);
		if (t < CH && (t & 63) == 0) L.wsum[t >> 6] = __popcll(wm);
		__syncthreads();
		const int nw0 = L.wsum[0], nwant = nw0 + L.wsum[1];
		if (HT_DBG(dbg, 0x200000) && t == 0) atomicAdd(nrows + b, nwant * 1000);      // tuning: how many points face away
		bool hit = false; v3 impact = V3(0, 0, 0);
		if (nwant > 0)
		{
			if (want) L.pair[((t >> 6) ? nw0 : 0) + __popcll(wm & ((1ull << (t & 63)) - 1ull))] = (unsigned short)(t | (rb << 8));
			__syncthreads();
			const int g = t & 15;
			auto row_min = [](int x) -> int {
				x = min(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false)); x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));
				x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false)); x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));
				return x;
			};
			auto row_or = [](int x) -> int {
				x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
				x |= __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); x |= __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);
				return x;
			};
			for (int q0 = 0; q0 < nwant; q0 += CR_THREADS / 16)
			{
				const int q = q0 + (t >> 4);
				const bool on = q < nwant;
				const unsigned e = on ? L.pair[q] : 0;
				const int pt = e & 255, body = e >> 8;
				const float *tr = tab + body * BT;
				const float4 pw = L.v[pt];
				v3 v0 = tab_to_local(tr, origin), v1 = tab_to_local(tr, V3(pw.x, pw.y, pw.z));
				const float4 *pl = s_planes + tab_plane0(tr);
				const int np = on ? tab_nplanes(tr) : 0;
				int k0 = 0; bool done = !on, ok = true;
				while (__any(!done))
				{
					// the next 16 faces, one per lane: which is the first that acts on the segment as it is now
					const int i = k0 + g;
					int first = 0x7fffffff; float e0 = 0.0f, e1 = 0.0f;
					if (!done && i < np)
					{
						const float4 f = pl[i];
						const v4 plane = V4(f.x, f.y, f.z, f.w);
						e0 = dot_plane(plane, v0); e1 = dot_plane(plane, v1);
						if ((e0 >= 0 && e1 >= 0) || !(e0 <= 0 && e1 <= 0)) first = i;
					}
					const int kmin = row_min(first);
					const bool mine = first == kmin && kmin != 0x7fffffff;      // one lane of the row at most
					const float d0 = __int_as_float(row_or(mine ? __float_as_int(e0) : 0)), d1 = __int_as_float(row_or(mine ? __float_as_int(e1) : 0));
					if (!done)
					{
						if (kmin == 0x7fffffff) { k0 += 16; if (k0 >= np) done = true; }      // none of the sixteen acts; past the last face the (clipped) segment lies inside
						else if (d0 >= 0 && d1 >= 0) { ok = false; done = true; }
						else
						{
							const v3 c = v0 + ((v1 - v0) * d0) / (d0 - d1);
							if (d0 >= 0) v0 = c; else v1 = c;
							k0 = kmin + 1;
							if (k0 >= np) done = true;
						}
					}
				}
				if (on && g == 0) { const v3 w = tab_to_world(tr, v0); L.v[pt] = make_float4(w.x, w.y, w.z, ok ? 1.0f : 0.0f); }
			}
			__syncthreads();
			if (want) { const float4 r = L.v[t]; hit = r.w != 0.0f; impact = V3(r.x, r.y, r.z); }
		}
		if (t >= CH) continue;                     // the other waves only help with the pair scans and the hit checks
		if (!active) continue;
		const float *tr = tab + rb * BT;
		v3 position1, normal;
		if (hit) { position1 = tab_to_local(tr, impact); normal = normalize(v - origin); }
		else { position1 = tab_to_local(tr, v - xyz(p) * dot_plane(p, v)); normal = xyz(p); }
		const float targetdist = dot(tab_to_world(tr, position1) - v, normal);                  // ConstrainAlongDirection physics.h:328-331
		float fmin = -1.0f, fmax = 1.0f;
		if (mode == 1) { float k = (rb == 0 || rb == 1 || rb == 2) ? weak_force : 1.0f; fmin = -1.0f * k * microforce; fmax = 1.0f * k * microforce; }
		else if (mode == 2) { float cloudforce = fmin_std(cf_max_point, cf_max_sum / (float)n); float k = (rb == 0) ? 0.1f : 1.0f; fmin = -cloudforce * k; fmax = cloudforce * k; }
		else if (mode == 3) { fmin = -1.0f * unibody_force; fmax = 1.0f * unibody_force; }
		else if (mode == 4) { const float k = (microforce * weak_force) * ((rb == 0) ? cf_max_point : 1.0f); fmin = -1.0f * k; fmax = 1.0f * k; }      // slowfit handtrack.h:815-816: weak_force = step ratio, cf_max_point = wrist factor
		if (rec.scratch)      // the solver's record of this row (ht_quad.hpp), at the point's index of the frame's scratch slot, and the row's body beside it
		{
			const v4 q = V4(wq[rb][0], wq[rb][1], wq[rb][2], wq[rb][3]);
			m3 I; I.x = V3(wI[rb][0], wI[rb][1], wI[rb][2]); I.y = V3(wI[rb][3], wI[rb][4], wI[rb][5]); I.z = V3(wI[rb][6], wI[rb][7], wI[rb][8]);
			const v3 r1 = qrot(q, position1);                                                     // LimitLinear::Iter physics.h:294
			const float impulsed = wI[rb][9] + dot(cross(mul(I, cross(r1, normal)), r1), normal);  // physics.h:299-300 with rb0 == NULL
			const float ts = targetdist / rec.dt;                                                  // PhysicsUpdate physics.h:553-554
			quad_write_record(rec.scratch + ((size_t)b * rec.stride + i) * HT_CREC, r1, normal, I, wI[rb][9], ts, fmin_std(ts, 0.0f), impulsed, fmin_std(fmin, fmax) * rec.dt, fmax_std(fmin, fmax) * rec.dt);
			rec.body[(size_t)b * M.pts_cap + i] = (unsigned char)rb;
			continue;
		}
		float4 *out = reinterpret_cast<float4 *>(rows + ((size_t)b * M.pts_cap + i) * HT_ROW);
		out[0] = make_float4(-1.0f, (float)rb, v.x, v.y);
		out[1] = make_float4(v.z, position1.x, position1.y, position1.z);
		out[2] = make_float4(normal.x, normal.y, normal.z, targetdist);
		out[3] = make_float4(0.0f, fmin_std(fmin, fmax), fmax_std(fmin, fmax), 0.0f);
	}
}
__global__ __launch_bounds__(CR_THREADS) void k_cloud_rows(ht_model_dev M, const float *__restrict__ state, const float4 *__restrict__ pts, const int *__restrict__ npts,
                                                           const float *__restrict__ cams, const int *__restrict__ active_flag, int stride, int use_cam_origin, int mode,
                                                           float microforce, float weak_force, float cf_max_point, float cf_max_sum, float unibody_force,
                                                           float *__restrict__ rows, int *__restrict__ nrows, cloud_records rec, int dbg)
{
	__shared__ float tab[HT_MAXNB * BT];
	__shared__ closest_lds L;
	__shared__ float wq[HT_MAXNB][4], wI[HT_MAXNB][10];      // record mode: the bodies' orientations, world inverse inertia and inverse mass, as k_solve forms them
	const int b = M.frame_order ? M.frame_order[blockIdx.x] : (int)blockIdx.x, t = threadIdx.x;
	const int n = npts[b];
	const int nsub = (n + stride - 1) / stride;
	if (active_flag && !active_flag[b]) return;      // a masked launch leaves the other frames' rows and counts alone (another launch may be producing them)
	if ((int)blockIdx.y * CH >= nsub && blockIdx.y > 0) return;      // gridDim.y blocks share a frame's passes (a row only depends on its own point)
	if (t == 0 && blockIdx.y == 0) nrows[b] = nsub;
	cloud_rows_frame(M, state, pts, cams, stride, use_cam_origin, mode, microforce, weak_force, cf_max_point, cf_max_sum, unibody_force, rows, nrows, rec, dbg, b, n, blockIdx.y, gridDim.y, tab, L, wq, wI);
}

// ------------------------------------------------------------------------------------------------- k_reset
// The full-reset branch of HandTracker::update (handtrack.h:706-711) for one flagged frame in ONE block: PoseFromScratch (handtrack.h:478-506), then
// steps_unibody times UnibodyFit (handtrack.h:451-470: the cloud rows of every 4th point, re-expressed on one proxy body, a single-body solve, the pose moved
// with it).  As seven launches of one-block-per-flagged-frame kernels the chain paid a cold start per launch (lone blocks: the planes, the body table and the
// code itself fetched again) while the batch waited for it; the phases share the block's LDS (planes + closest-feature scratch | the solve's records).
// Two builds: MINB 1 takes the registers the chain wants (297: one block per CU) and is the one an update launches while few frames reset -- the batch waits
// for the slowest of them (8 of 1024 frames: 0.48 ms against 0.53); MINB 2 (256 registers, two blocks per CU) serves a launch over more frames than the device
// has CUs (all 1024 frames: 1.13 ms against 2.08; tools/reset_all_frames.py).  The context keeps count of the frames that reset (ht_host.hpp: d_nreset).
#define RS_THREADS CR_THREADS
__device__ __forceinline__ v3 G3(const float *p) { return V3(p[0], p[1], p[2]); }
__device__ __forceinline__ v4 G4(const float *p) { return V4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ m3 GM(const float *p) { m3 m; m.x = V3(p[0], p[1], p[2]); m.y = V3(p[3], p[4], p[5]); m.z = V3(p[6], p[7], p[8]); return m; }
#define UB_LDS_ROWS 896      // rows of the single-body solve kept in LDS (3584 points); 72 KB with sums and chain: two blocks per CU
#define UB16_ROWS 448        // rows up to which the solve runs sixteen rows at a time (ht_quad.hpp: quad_blocks16_run): records + couplings + sums of 448 + 32 rows are 63 KB of the same 66
static_assert((size_t)(UB16_ROWS + QUAD_B16_SLACK) * (CREC + 16 + 1) * sizeof(float) <= (size_t)(UB_LDS_ROWS + QUAD_CHAIN_SLACK) * (CREC * sizeof(float) + sizeof(float) + sizeof(unsigned short)), "the blocked solve's arrays must fit the row-by-row solve's LDS");
template <bool EXACT> __device__ __forceinline__ void reset_frame(const ht_model_dev &M, const ht_physics_dev &ph, float *state, const float4 *__restrict__ pts, const int *__restrict__ npts,
                                            const float *__restrict__ analysis, const float *__restrict__ cams, int n_unibody,
                                            float unibody_force, float *rows, int *nrows, float *scratch, int scratch_stride, int batch, int dbg, const int b)
{
	__shared__ float pos[HT_MAXNB][3], q[HT_MAXNB][4];
	__shared__ float pc[3], res[8];
	__shared__ float fj[HT_MAXNJ][8], fc[HT_MAXNJ][6];
	const int t = threadIdx.x;
	const int nb = M.nb;
	const int n = npts[b];
	float *st = state + (size_t)b * nb * HT_STATE_STRIDE;
	const float *an = analysis + (size_t)b * HT_ANALYSIS;
	const float *cam = cams + (size_t)b * HT_CAM;
	// the dynamic LDS, phase by phase
	float4 *const wp = s_planes;                                                                        // PoseFromScratch: 256 weighted points
	closest_lds &L = *reinterpret_cast<closest_lds *>(s_planes + ((M.plane_off[nb] + 16 + 3) & ~3));      // cloud rows: planes, closest-feature scratch, body table
	float *const tab = reinterpret_cast<float *>(reinterpret_cast<char *>(&L) + ((sizeof(closest_lds) + 15) & ~15));
	float *const urow = reinterpret_cast<float *>(s_planes);                                            // single-body solve: records, impulse sums, chain
	float *const usum = urow + (UB_LDS_ROWS + QUAD_CHAIN_SLACK) * CREC;
	unsigned short *const uidx = reinterpret_cast<unsigned short *>(usum + UB_LDS_ROWS + QUAD_CHAIN_SLACK);
	float *const ug = urow + (UB16_ROWS + QUAD_B16_SLACK) * CREC;                                       // the same, sixteen rows at a time (up to UB16_ROWS rows): records, couplings, impulse sums
	float *const usum16 = ug + (UB16_ROWS + QUAD_B16_SLACK) * 16;

	{
		// ---- PoseFromScratch.  Palm ray from the first three landmark rays, inverse-distance weighted centroid of the cloud (handtrack.h:483-490): the weights are
		// independent per point (256 per pass, one per thread, through LDS); the sums keep the reference's order on thread 0, eight terms read ahead.
		const v4 cs = (G4(an + HT_AN_CRAYS) + G4(an + HT_AN_CRAYS + 4)) + G4(an + HT_AN_CRAYS + 8);
		const v3 palmray = normalize(xyz(cs));
		v3 pcom = V3(0, 0, 0); float wsum = 0.00000000001f;
		for (int base = 0; base < n; base += RS_THREADS)
		{
			const int m = min(RS_THREADS, n - base);
			__syncthreads();
			if (t < m)
			{
				const float4 pv = pts[(size_t)b * M.pts_cap + base + t];
				const v3 p = V3(pv.x, pv.y, pv.z);
				const v3 c = cross(p, palmray);
				const float w = 1.0f / (0.000001f + dot(c, c));
				const v3 pw = p * w;
				wp[t] = make_float4(pw.x, pw.y, pw.z, w);
			}
			__syncthreads();
			if (t == 0)
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
		if (t == 0) { pcom = pcom / wsum; pc[0] = pcom.x; pc[1] = pcom.y; pc[2] = pcom.z; }
		if (t < nb)
		{
			const float *bc = M.bodyc + t * HT_BC;     // Reset(rb) physmodel.h:221-226
			for (int i = 0; i < 3; i++) pos[t][i] = bc[HT_BC_POS0 + i];
			for (int i = 0; i < 4; i++) q[t][i] = bc[HT_BC_Q0 + i];
		}
		if (t >= 64 && t < 64 + M.nj)      // the joints' constants for FixPositions below
		{
			const int j = t - 64;
			const float *jc = M.jointc + j * HT_JC;
			const int r0 = (int)jc[HT_JC_RB0], r1 = (int)jc[HT_JC_RB1];
			fj[j][0] = (float)r0; fj[j][1] = (float)r1;
			for (int i = 0; i < 3; i++) { fj[j][2 + i] = jc[HT_JC_P0 + i]; fj[j][5 + i] = jc[HT_JC_P1 + i]; }
			for (int i = 0; i < 3; i++) { fc[j][i] = M.bodyc[r0 * HT_BC + HT_BC_COM + i]; fc[j][3 + i] = M.bodyc[r1 * HT_BC + HT_BC_COM + i]; }
		}
		__syncthreads();
		const v4 camq = V4(cam[8], cam[9], cam[10], cam[11]);
		const v4 palmq = G4(an + HT_AN_PALMQ);
		const xf p1 = XF(V3(pc[0], pc[1], pc[2]), qmul(camq, palmq));
		const xf dp = mul(p1, inverse(XF(G3(pos[1]), G4(q[1]))));
		__syncthreads();
		if (t < nb)
		{
			const xf np = mul(dp, XF(G3(pos[t]), G4(q[t])));
			pos[t][0] = np.p.x; pos[t][1] = np.p.y; pos[t][2] = np.p.z; q[t][0] = np.q.x; q[t][1] = np.q.y; q[t][2] = np.q.z; q[t][3] = np.q.w;
		}
		__syncthreads();
		if (t >= 1 && t <= 4 && nb >= 17)     // curl the four fingers by the decoded clench angles (handtrack.h:498-504)
		{
			const int finger = t;
			const float a = an[HT_AN_CLENCH + finger];
			const v4 jf = G4(M.jointc + (1 + finger * 3) * HT_JC + HT_JC_FRAME);
			const float ang[3] = { a / 2.0f, a, a * 1.25f };
			for (int k = 0; k < 3; k++)
			{
				const int bb = 2 + k + finger * 3;
				const v4 o = qmul(jf, qmul(G4(q[bb]), quat_axis_angle(V3(1, 0, 0), ang[k])));
				q[bb][0] = o.x; q[bb][1] = o.y; q[bb][2] = o.z; q[bb][3] = o.w;
			}
		}
		__syncthreads();
		if (t == 0)      // FixPositions: ordered top-down (physmodel.h:404-408)
		{
			for (int j = 0; j < M.nj; j++)
			{
				const int r0 = (int)fj[j][0], r1 = (int)fj[j][1];
				const xf u0 = XF(apply(XF(G3(pos[r0]), G4(q[r0])), -G3(fc[j])), G4(q[r0]));
				const xf u1 = XF(apply(XF(G3(pos[r1]), G4(q[r1])), -G3(fc[j] + 3)), G4(q[r1]));
				const v3 np = G3(pos[r1]) + (apply(u0, G3(fj[j] + 2)) - apply(u1, G3(fj[j] + 5)));
				pos[r1][0] = np.x; pos[r1][1] = np.y; pos[r1][2] = np.z;
			}
		}
		__syncthreads();
		if (t < nb)
		{
			float *s = st + t * HT_STATE_STRIDE;
			for (int i = 0; i < 3; i++) s[i] = pos[t][i];
			for (int i = 0; i < 4; i++) s[3 + i] = q[t][i];
			for (int i = 7; i < 13; i++) s[i] = 0.0f;
		}
		__syncthreads();
	}

	const int nsub = (n + 3) / 4;
	if (t == 0 && n_unibody > 0) nrows[b] = nsub;
	const cloud_records none = { nullptr, 0, nullptr, 0.0f };
	for (int it = 0; it < n_unibody; it++)
	{
		// ---- the cloud rows of every 4th point from the camera's origin, UnibodyFit's force limits (handtrack.h:457-461)
		cloud_rows_frame(M, state, pts, cams, 4, 1, 3, 0.0f, 0.0f, 0.0f, 0.0f, unibody_force, rows, nrows, none, dbg, b, n, 0, 1, tab, L, nullptr, nullptr);
		__syncthreads();
		// ---- UnibodyFit's solve: all rows act on one proxy body, so the Gauss-Seidel chain is sequential: a wave takes it sixteen rows at a time (ht_quad.hpp: the rows' velocity
		//      terms side by side, the impulses resolved in row order through pre-computed couplings); a cloud too large for that, one quad row by row
		if (t < nb) { for (int i = 0; i < 3; i++) pos[t][i] = st[t * HT_STATE_STRIDE + i]; for (int i = 0; i < 4; i++) q[t][i] = st[t * HT_STATE_STRIDE + 3 + i]; }
		__syncthreads();
		// SanityCheck before the solve (handtrack.h:463) is a no-op unless the pose already holds NaNs; those bodies are reset
		if (t < nb)
		{
			bool bad = false;
			for (int i = 0; i < 13; i++) bad = bad || isnan(st[t * HT_STATE_STRIDE + i]);
			if (bad) { for (int i = 0; i < 3; i++) pos[t][i] = M.bodyc[t * HT_BC + HT_BC_POS0 + i]; for (int i = 0; i < 4; i++) q[t][i] = M.bodyc[t * HT_BC + HT_BC_Q0 + i]; }
		}
		__syncthreads();
		const float dt = ph.deltaT;
		const v3 ubpos = G3(pos[1]) + V3(M.ub_com[0], M.ub_com[1], M.ub_com[2]);        // RigidBody ctor: position += com (physics.h:157)
		const v4 ubq = G4(q[1]);
		const xf ubi = inverse(XF(ubpos, ubq));
		const float minv = M.ub_massinv;
		const m3 tinv = GM(M.ub_tinv);
		const m3 Iinv = world_inertia(ubq, tinv, minv);
		// The rows re-expressed on the proxy body and pre-computed into records (handtrack.h:457-462).  Up to UB_LDS_ROWS rows the records stay in LDS (a single
		// quad walking its chain alone on a CU would wait a whole L2 round trip for what k_solve's sixteen quads overlap); a larger cloud uses the frame's slot
		// of the solver scratch in HBM, sums behind all frames' records as in k_solve.
		const int nr = nsub < scratch_stride - QUAD_CHAIN_SLACK ? nsub : scratch_stride - QUAD_CHAIN_SLACK;
		if constexpr (EXACT)
		{
			// tests only (ht_debug_solver_build 5): the reference's own sweeps over the rows as UnibodyFit re-expresses them (handtrack.h:457-462), LimitLinear::Iter
			// (physics.h:289-307) row by row on one lane, no fused multiply-adds
			float *const gs = scratch + (size_t)batch * scratch_stride * CREC + (size_t)b * scratch_stride;
			for (int i = t; i < nr; i += RS_THREADS)
			{
				float *r = rows + ((size_t)b * M.pts_cap + i) * HT_ROW;
				const int rb1 = (int)r[1];
				const v3 p1 = apply(ubi, apply(XF(G3(pos[rb1]), G4(q[rb1])), G3(r + 5)));
				r[5] = p1.x; r[6] = p1.y; r[7] = p1.z;
				gs[i] = 0.0f;
			}
			__threadfence_block();
			__syncthreads();
			if (t == 0)
			{
				v3 lin = V3((0.0f * M.ub_dampleft) + 0.0f, (0.0f * M.ub_dampleft) + 0.0f, (0.0f * M.ub_dampleft) + 0.0f), ang = lin;
				v3 pn = ubpos; v4 qn = ubq;
				const int total = ph.iterations + ph.iterations_post;
				for (int sweep = 0; sweep < total; sweep++)
				{
					const bool post = sweep >= ph.iterations;
					for (int i = 0; i < nr; i++)
					{
						const float *r = rows + ((size_t)b * M.pts_cap + i) * HT_ROW;
						const v3 p0 = G3(r + 2), p1 = G3(r + 5), nrm = G3(r + 8);
						const float ts0 = r[11] / dt, ts = post ? fmin_std(ts0, r[12]) : ts0;
						const v3 r1 = qrot(ubq, p1);
						const v3 v0 = V3(0, 0, 0), v1 = cross(mul(Iinv, ang), r1) + lin * minv;
						const float vn = dot(v1 - v0, nrm);
						const float impulsen = -ts - vn;
						const float impulsed = 0.0f + (minv + dot(cross(mul(Iinv, cross(r1, nrm)), r1), nrm));
						float impulse = impulsen / impulsed;
						const float isum = gs[i];
						impulse = fmin_std(r[14] * dt - isum, impulse);
						impulse = fmax_std(r[13] * dt - isum, impulse);
						const v3 im = nrm * impulse;
						lin = lin + im; ang = ang + cross(r1, im);
						gs[i] = isum + impulse;
						(void)p0;
					}
					if (sweep + 1 == ph.iterations)
					{
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
				res[0] = pn.x; res[1] = pn.y; res[2] = pn.z; res[3] = qn.x; res[4] = qn.y; res[5] = qn.z; res[6] = qn.w;
			}
		}
		else
		{
		const bool in_lds = nr <= UB_LDS_ROWS;
		const bool blk16 = nr <= UB16_ROWS && !HT_DBG(dbg, 8192);      // sixteen rows at a time (ht_quad.hpp); a larger cloud row by row
		const int nblk16 = (nr + 15) >> 4;
		float *const grec = scratch + (size_t)b * scratch_stride * CREC;
		float *const gsum = scratch + (size_t)batch * scratch_stride * CREC + (size_t)b * scratch_stride;
		unsigned *const gidx = reinterpret_cast<unsigned *>(scratch + (size_t)batch * scratch_stride * (CREC + 1)) + (size_t)b * scratch_stride;
		if (blk16) { for (int i = t; i < 16 * nblk16 + QUAD_B16_SLACK; i += RS_THREADS) { usum16[i] = 0.0f; if (i >= nr) quad_write_noop(urow + (size_t)i * CREC); } }      // the last block is filled up with rows that change nothing
		else if (in_lds) { for (int i = t; i < nr + QUAD_CHAIN_SLACK; i += RS_THREADS) { usum[i] = 0.0f; uidx[i] = (unsigned short)i; } }
		else for (int i = t; i < nr + QUAD_CHAIN_SLACK; i += RS_THREADS) { gsum[i] = 0.0f; gidx[i] = (unsigned)i; }
		for (int i = t; i < nr; i += RS_THREADS)
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
		if (blk16)      // the couplings of every row with the rows before it in its block of sixteen: one float4 (four couplings) per entry
		{
			const float4 *const r4 = reinterpret_cast<const float4 *>(urow);
			for (int e = t; e < 64 * nblk16; e += RS_THREADS)
			{
				const int r = e >> 2, k = e & 3, j = r & 15, first = r & ~15;
				float v[4];
#pragma unroll
				for (int u = 0; u < 4; u++) { const int i = 4 * k + u; v[u] = i < j ? quad_coupling16(r4 + 4 * r, r4 + 4 * (first + i)) : 0.0f; }
				reinterpret_cast<float4 *>(ug)[e] = make_float4(v[0], v[1], v[2], v[3]);
			}
			__syncthreads();
		}
		if (blk16 ? t < 64 : t < 4)       // the proxy body in quad layout (ht_quad.hpp): lane c < 3 of a quad owns component c, lane 3 carries the row's target speed; one quad walks the rows, or a wave's sixteen quads sixteen rows at a time
		{
			const int c = t & 3;
			// rbinitvelocity on a body at rest: 0 * damping + 0
			quad_body qb = { (0.0f * M.ub_dampleft) + 0.0f, (0.0f * M.ub_dampleft) + 0.0f };
			v3 pn = ubpos; v4 qn = ubq;
			const int total = ph.iterations + ph.iterations_post;
			for (int sweep = 0; sweep < total; sweep++)
			{
				const int tsoff = sweep >= ph.iterations ? 1 : 0;        // RemoveBias: lane 3 switches to the ts_post slot
				if (nr > 0) { if (blk16) quad_blocks16_run(qb, urow, ug, usum16, nblk16, t, tsoff); else if (in_lds) quad_chain_run(qb, urow, uidx, usum, nr, c, tsoff); else quad_chain_run(qb, grec, gidx, gsum, nr, c, tsoff); }
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
			if (t == 0) { res[0] = pn.x; res[1] = pn.y; res[2] = pn.z; res[3] = qn.x; res[4] = qn.y; res[5] = qn.z; res[6] = qn.w; }
		}
		}      // !EXACT
		__syncthreads();
		const xf dp = mul(XF(V3(res[0], res[1], res[2]), V4(res[3], res[4], res[5], res[6])), inverse(XF(G3(pos[1]), G4(q[1]))));
		if (t < nb)
		{
			xf np = mul(dp, XF(G3(pos[t]), G4(q[t])));
			float *s = st + t * HT_STATE_STRIDE;
			bool bad = isnan(np.p.x) || isnan(np.p.y) || isnan(np.p.z) || isnan(np.q.x) || isnan(np.q.y) || isnan(np.q.z) || isnan(np.q.w);
			for (int i = 7; i < 13; i++) bad = bad || isnan(s[i]);
			if (bad) { const float *bc = M.bodyc + t * HT_BC; np = XF(G3(bc + HT_BC_POS0), G4(bc + HT_BC_Q0)); for (int i = 7; i < 13; i++) s[i] = 0.0f; }
			s[0] = np.p.x; s[1] = np.p.y; s[2] = np.p.z; s[3] = np.q.x; s[4] = np.q.y; s[5] = np.q.z; s[6] = np.q.w;
		}
		__syncthreads();      // the next round's body table reads the pose; its planes overwrite the records
	}
}
// list != null: the frames list[0 .. *nlist) (the decision kernel's list of flagged frames, in no particular order); otherwise all B frames.  A block takes
// every gridDim.x-th entry: the grid is as large as the device holds blocks at once, not as large as the batch -- a block that finds nothing to do costs a
// launch of four waves all the same (and this kernel's waves start slowly: 1024 idle blocks cost 70 us, 8192 cost 600).
template <int MINB, bool EXACT = false> __global__ __launch_bounds__(RS_THREADS, MINB) void k_reset(ht_model_dev M, ht_physics_dev ph, float *state, const float4 *__restrict__ pts, const int *__restrict__ npts,
                                                                             const float *__restrict__ analysis, const float *__restrict__ cams, const int *__restrict__ list, const int *__restrict__ nlist, int B,
                                                                             int n_unibody, float unibody_force, float *rows, int *nrows, float *scratch, int scratch_stride, int batch, int dbg)
{
	const int n = list ? min(*nlist, B) : B;
	for (int slot = blockIdx.x; slot < n; slot += gridDim.x)
	{
		reset_frame<EXACT>(M, ph, state, pts, npts, analysis, cams, n_unibody, unibody_force, rows, nrows, scratch, scratch_stride, batch, dbg, list ? list[slot] : slot);
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------- k_fit_error
#ifdef HT_TUNING
__constant__ int ht_fit_error_dbg;
#endif
__global__ __launch_bounds__(256) void k_fit_error(ht_model_dev M, const float *__restrict__ state, const float4 *__restrict__ pts, const int *__restrict__ npts,
                                                  const uint16_t *__restrict__ depth, const float *__restrict__ cams, int w, int h, float bone_sum_error_scale, float *__restrict__ err, ht_fit_after after)
{
	__shared__ float tab[HT_MAXNB * BT];
	__shared__ closest_lds L;
	__shared__ int perr[HT_MAXNB];
	__shared__ int s_take;
	const int b = M.frame_order ? M.frame_order[blockIdx.x] : (int)blockIdx.x, t = threadIdx.x;
	if (t < 64) body_table_build(M, state + (size_t)b * M.nb * HT_STATE_STRIDE, tab, t);
	if (t < HT_MAXNB) perr[t] = 0;
	stage_planes(M, t, 256);
	__syncthreads();
	const int n = npts[b];
	int npmax = 0;
	for (int bb = 0; bb < M.nb; bb++) npmax = max(npmax, M.plane_off[bb + 1] - M.plane_off[bb]);      // the model's largest face count
#ifdef HT_TUNING
	long long cyc[6] = { 0, 0, 0, 0, 0, 0 }; const bool stats = (ht_fit_error_dbg & 0x400000) != 0; const long long t_in = stats ? clock64() : 0;
#endif
	for (int base = 0; base < n; base += CH)
	{
		const int i = base + (t & (CH - 1));
		const bool exists = i < n, active = t < CH && exists;
		const float4 pv = exists ? pts[(size_t)b * M.pts_cap + i] : make_float4(0, 0, 0, 0);
		int rb; v4 p; float dmin;
#ifdef HT_TUNING
		closest_chunk<256>(M, tab, L, active, exists, V3(pv.x, pv.y, pv.z), npmax, rb, p, dmin, stats ? cyc : nullptr);
#else
		closest_chunk<256>(M, tab, L, active, exists, V3(pv.x, pv.y, pv.z), npmax, rb, p, dmin);
#endif
		// pointerror[bone] = max(pointerror[bone], d) with pointerror starting at 0 (handtrack.h:376-383): only d > 0 matters,
		// and for non-negative floats the integer order of the bit patterns is the float order
		if (active && rb >= 0 && dmin > 0.0f) atomicMax(&perr[rb], __float_as_int(dmin));
	}
	__syncthreads();
#ifdef HT_TUNING
	if (stats && (b & 127) == 0 && (t == 0 || t == 192)) printf("fit_error frame %d thread %d: points %d pairs %lld; cycles A %lld prefix %lld B %lld C %lld, whole kernel %lld\n", b, t, n, cyc[5], cyc[0], cyc[1], cyc[2], cyc[3], (long long)(clock64() - t_in));
#endif
	if (t == 0)
	{
		float point_error_sum = 0.0f;
		for (int k = 0; k < M.nb; k++) point_error_sum += __int_as_float(perr[k]);
		const float *cam = cams + (size_t)b * HT_CAM;
		xf ci = inverse(XF(V3(cam[5], cam[6], cam[7]), V4(cam[8], cam[9], cam[10], cam[11])));
		float bone_error_sum = 0;
		for (int k = 0; k < M.nb; k++)
		{
			v3 position = apply(ci, tab_pos(tab + k * BT));
			const float fpx = position.x / position.z * cam[0] + cam[2], fpy = position.y / position.z * cam[1] + cam[3];     // projectz misc_image.h:50
			// int2(float2): a NaN (a body whose carried pose is NaN) converts to INT_MIN in the compiled reference and falls out of the image; the device's conversion
			// gives 0, which would be pixel (0, 0)
			const int px = fpx != fpx ? INT_MIN : (int)fpx, py = fpy != fpy ? INT_MIN : (int)fpy;
			if (!(px >= 0 && px <= w - 1 && py >= 0 && py <= h - 1)) continue;
			float bone_error = (float)(int)depth[((size_t)b * h + py) * w + px] * cam[4] - position.z;
			bone_error_sum += clamp_std(bone_error, 0.0f, 0.01f);
		}
		const float e = point_error_sum + bone_error_sum * bone_sum_error_scale;
		err[b] = e;
		if (after.mode == 1) { const int f = (after.angles_only || e > after.reset_thr) ? 1 : 0; after.flags[b] = f; after.nflags[b] = !f; if (f && after.list) { const int k = atomicAdd(after.nlist, 1); if (k < (int)gridDim.x) after.list[k] = b; } if (after.nreset) { if (f) atomicAdd(after.nreset, 1u); if (b == 0) atomicAdd(after.nreset + 1, 1u); } }      // handtrack.h:706
		if (after.mode == 2)      // handtrack.h:713-731: the CNN-driven pose replaces the tracked one when it explains the frame better for long enough
		{
			float pfe = after.prev_err[b];
			const float olderror = after.err_old[b], newerror = e;
			if (newerror > olderror) pfe = 0.0f; else pfe += olderror - newerror;
			const bool take = (n > after.min_point_num && after.initializing[b]) || after.always_take_cnn || after.angles_only || pfe > after.accum_thr;
			if (pfe > after.accum_thr) pfe = 0.0f;
			after.prev_err[b] = pfe;
			const int ini = after.initializing[b] - 1;
			after.initializing[b] = ini < 0 ? 0 : ini;
			if (after.accepted) after.accepted[b] = take ? after.nb : 0;
			s_take = take ? 1 : 0;
		}
	}
	if (after.mode == 2)
	{
		__syncthreads();
		if (s_take && after.hand)      // handmodel.SetPose(othermodel pose): the momenta stay
			for (int i = t; i < after.nb * 7; i += 256) { const size_t o = ((size_t)b * after.nb + i / 7) * HT_STATE_STRIDE + i % 7; after.hand[o] = after.other[o]; }
	}
}

// ------------------------------------------------------------------------------------------------- k_chamber
// 5 silhouette planes through the camera origin + one ConstrainUnderPlane row per (plane, body); rows [B][5*nb][HT_ROW], nch[b] = 0 or 5*nb.
// planes_in / on_in (round 6): the frame's five planes as k_chamber_planes below left them -- they follow from the points alone, so the three main-thread passes of an
// update (handtrack.h:769-780) share one scan made beside the net; null: the scan is made here (a pass outside an update: ht_stage_chamber, ht_stage_fit).
// Four waves per frame share the (plane, body) items: the kernel is the items' memory round trips one after the other, 85 items on one wave were 69 us per launch.
#define CHB_THREADS 256
__global__ __launch_bounds__(CHB_THREADS) void k_chamber(ht_model_dev M, const float *__restrict__ state, const float4 *__restrict__ pts, const int *__restrict__ npts,
                                                        int min_point_num, int enabled, float maxforce, float *__restrict__ rows, int *__restrict__ nch,
                                                        const float *__restrict__ planes_in, const int *__restrict__ on_in)
{
	__shared__ float tab[HT_MAXNB * BT];
	__shared__ float planes[5][4];
	__shared__ int voff[HT_MAXNB + 1];      // vertex ranges of the bodies (picked per lane below: from LDS, not from the kernel-argument segment)
	__shared__ float4 chunk[256];
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int n = npts[b];
	const bool on = planes_in ? on_in[b] != 0 : (enabled && n > min_point_num);        // handtrack.h:774
	if (t == 0) nch[b] = on ? 5 * M.nb : 0;
	if (!on) return;
	if (wave == 0) body_table_build(M, state + (size_t)b * M.nb * HT_STATE_STRIDE, tab, lane);
	if (wave == 1 && lane <= M.nb) voff[lane] = M.vert_off[lane];
	if (planes_in)
	{
		if (wave == 2 && lane < 20) planes[lane >> 2][lane & 3] = planes_in[(size_t)b * 20 + lane];
	}
	else
	{
		// containing_plane (physmodel.h:183-193): a sequential scan of the cloud per direction (the running `best` decides the next comparison).  The five
		// directions take one lane each; the points reach them through an LDS chunk the block fills with coalesced reads (a lane reading the
		// points straight from HBM waits a memory round trip per point: 100 us for a 424-point frame).
		const float od[5][3] = { { -1, -0.25f, 0 }, { -1, -1, 0 }, { 0, -1, 0 }, { 1, -1, 0 }, { 1, -0.25f, 0 } };    // handtrack.h:776
		const int dl = t < 5 ? t : 0;
		const v3 outdir = V3(od[dl][0], od[dl][1], od[dl][2]), origin = V3(0, 0, 0), viewdir = V3(0, 0, 1);
		v3 best = viewdir - outdir;
		best = best + origin;
		const v3 tangent = cross(best, outdir);
		for (int base = 0; base < n; base += 256)
		{
			const int m = min(256, n - base);
			__syncthreads();
			if (t < m) chunk[t] = pts[(size_t)b * M.pts_cap + base + t];
			__syncthreads();
			if (t < 5)
				for (int i = 0; i < m; i++)
				{
					const float4 pv = chunk[i];
					const v3 p = V3(pv.x, pv.y, pv.z);
					if (dot(cross(best - origin, p - origin), tangent) > 0) best = p;
				}
		}
		if (t < 5)
		{
			v3 nn = normalize(cross(tangent, best));
			planes[t][0] = nn.x; planes[t][1] = nn.y; planes[t][2] = nn.z; planes[t][3] = -dot(nn, origin);
		}
	}
	__syncthreads();
	// one ConstrainUnderPlane row (physics.h:347-350) per (plane, body): the body's support vertex against the plane normal (maxdir geometric.h:218-224: the
	// FIRST maximum in vertex order).  A DPP row of sixteen lanes takes an item and its vertices sixteen at a time, four reads ahead (one lane walking a
	// body's ~180 vertices in global memory waited a memory round trip per vertex: 76 of the kernel's 140 us); the sixteen first-maxima merge under
	// "larger value, then lower index", which is the sequential scan's winner.
	const int g = lane & 15;
	for (int item0 = 4 * wave; item0 < 5 * M.nb; item0 += 4 * (CHB_THREADS / 64))
	{
		const int item = item0 + (lane >> 4);
		const bool on = item < 5 * M.nb;
		const int d = on ? item / M.nb : 0, rb = on ? item % M.nb : 0;
		const float *t = tab + rb * BT;
		v4 plane = V4(planes[d][0], planes[d][1], planes[d][2], planes[d][3]);
		v3 dirl = qrot(qconj(V4(t[3], t[4], t[5], t[6])), xyz(plane));
		const float4 *vs = M.verts + voff[rb];
		const int nv = on ? voff[rb + 1] - voff[rb] : 0;
		int bi = 0x7fffffff; float bd = 0.0f;
		for (int k = g; k < nv; k += 64)
		{
			const int k1 = k + 16, k2 = k + 32, k3 = k + 48;
			const float4 q0 = vs[k], q1 = vs[k1 < nv ? k1 : k], q2 = vs[k2 < nv ? k2 : k], q3 = vs[k3 < nv ? k3 : k];
			const float d0 = dot(V3(q0.x, q0.y, q0.z), dirl), d1 = dot(V3(q1.x, q1.y, q1.z), dirl), d2 = dot(V3(q2.x, q2.y, q2.z), dirl), d3 = dot(V3(q3.x, q3.y, q3.z), dirl);
			if (bi == 0x7fffffff || bd < d0) { bd = d0; bi = k; }
			if (k1 < nv && bd < d1) { bd = d1; bi = k1; }
			if (k2 < nv && bd < d2) { bd = d2; bi = k2; }
			if (k3 < nv && bd < d3) { bd = d3; bi = k3; }
		}
		// merge over the row: larger value wins, equal values go to the lower vertex index, "nothing" (a lane without a vertex) loses
#pragma unroll
		for (int st = 0; st < 4; st++)
		{
			float ob; int oi;
			if (st == 0) { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0xB1, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0xB1, 0xF, 0xF, false); }
			else if (st == 1) { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x4E, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0x4E, 0xF, 0xF, false); }
			else if (st == 2) { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x141, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0x141, 0xF, 0xF, false); }
			else { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x140, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0x140, 0xF, 0xF, false); }
			const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || bd < ob || (ob == bd && oi < bi));
			bd = take ? ob : bd; bi = take ? oi : bi;
		}
		if (on && g == 0)
		{
			float4 q = vs[bi];
			v3 sv = V3(q.x, q.y, q.z);
			v3 p0 = xyz(plane) * -plane.w, axis = -xyz(plane);
			float targetdist = dot(tab_to_world(t, sv) - p0, axis);
			float4 *out = reinterpret_cast<float4 *>(rows + ((size_t)b * 5 * M.nb + item) * HT_ROW);
			out[0] = make_float4(-1.0f, (float)rb, p0.x, p0.y);
			out[1] = make_float4(p0.z, sv.x, sv.y, sv.z);
			out[2] = make_float4(axis.x, axis.y, axis.z, targetdist);
			out[3] = make_float4(0.0f, fmin_std(0.0f, maxforce), fmax_std(0.0f, maxforce), 0.0f);
		}
	}
}

// ------------------------------------------------------------------------------------------------- k_chamber_planes
// The five containing planes alone (physmodel.h:183-193): they follow from the frame's points, not from the pose, so the three main-thread passes of an update
// (handtrack.h:769-780) share them -- one scan per update, beside the net, instead of one per pass; the planes' rows (one per plane and body) are made pass by pass
// by k_solve_prep (csrc/ht_prep.hip).  planes [B][5][4], on [B] = the frame has them (handtrack.h:774).  Same statements as k_chamber's first half.
__global__ __launch_bounds__(64) void k_chamber_planes(ht_model_dev M, const float4 *__restrict__ pts, const int *__restrict__ npts, int min_point_num, int enabled, float *__restrict__ planes, int *__restrict__ onflag)
{
	__shared__ float4 chunk[256];
	const int b = blockIdx.x, lane = threadIdx.x;
	const int n = npts[b];
	const bool on = enabled && n > min_point_num;        // handtrack.h:774
	if (lane == 0) onflag[b] = on ? 1 : 0;
	if (!on) return;
	const float od[5][3] = { { -1, -0.25f, 0 }, { -1, -1, 0 }, { 0, -1, 0 }, { 1, -1, 0 }, { 1, -0.25f, 0 } };    // handtrack.h:776
	const int dl = lane < 5 ? lane : 0;
	const v3 outdir = V3(od[dl][0], od[dl][1], od[dl][2]), origin = V3(0, 0, 0), viewdir = V3(0, 0, 1);
	v3 best = viewdir - outdir;
	best = best + origin;
	const v3 tangent = cross(best, outdir);
	for (int base = 0; base < n; base += 256)
	{
		const int m = min(256, n - base);
		__syncthreads();
		for (int i = lane; i < m; i += 64) chunk[i] = pts[(size_t)b * M.pts_cap + base + i];
		__syncthreads();
		if (lane < 5)
			for (int i = 0; i < m; i++)
			{
				const float4 pv = chunk[i];
				const v3 p = V3(pv.x, pv.y, pv.z);
				if (dot(cross(best - origin, p - origin), tangent) > 0) best = p;
			}
	}
	if (lane < 5)
	{
		v3 nn = normalize(cross(tangent, best));
		float *o = planes + (size_t)b * 20 + 4 * lane;
		o[0] = nn.x; o[1] = nn.y; o[2] = nn.z; o[3] = -dot(nn, origin);
	}
}
void ht_launch_chamber_planes(const ht_model_dev &M, const float4 *pts, const int *npts, int min_point_num, int enabled, float *planes, int *on, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_chamber_planes, dim3(B), dim3(64), 0, s, M, pts, npts, min_point_num, enabled, planes, on);
}

// ------------------------------------------------------------------------------------------------- launchers
// `rec`: instead of the 16-float rows, write each row's solver record into the frames' scratch slots and the rows' bodies into rec->body (k_solve then only
// lists them per body); null: the reference-layout rows (stage calls, UnibodyFit)
void ht_launch_cloud_rows(const ht_model_dev &M, const float *state, const float4 *pts, const int *npts, const float *cams, const int *active_flag, int stride, int use_cam_origin, int mode,
                          const ht_params &par, float *rows, int *nrows, int B, hipStream_t s, float sf_ratio, float sf_wrist, const cloud_records *rec)
{
	const cloud_records none = { nullptr, 0, nullptr, 0.0f };
	// blocks per frame: a frame's passes of CH points are independent, so while the batch leaves CUs idle they are spread over up to `split` blocks
	// (each pays the prologue -- body table, 25 KB of planes into LDS -- again, which is why a large batch keeps one block per frame)
	const int pts_max = M.pts_bound > 0 ? M.pts_bound : M.pts_cap, passes = ((pts_max + stride - 1) / stride + CH - 1) / CH;
	int split = B <= 2048 ? 2 : 1;
#ifdef HT_TUNING
	if (const char *e = getenv("HT_CLOUD_SPLIT")) split = atoi(e);
#endif
	if (split > passes) split = passes;
	if (split < 1) split = 1;
	hipLaunchKernelGGL(k_cloud_rows, dim3(B, split), dim3(CR_THREADS), ((size_t)M.plane_off[M.nb] + 16) * sizeof(float4), s, M, state, pts, npts, cams, active_flag, stride, use_cam_origin, mode, par.microforce,
	                   mode == 4 ? sf_ratio : par.physics_weak_force, mode == 4 ? sf_wrist : par.cloudforce_max_point, par.cloudforce_max_sum, par.unibody_force, rows, nrows, rec ? *rec : none, ht_tuning_flags());
}
// the full-reset branch for the frames list[0 .. *nlist) (all B frames with list == nullptr); many_frames: the caller expects more of them than the device has CUs
// (the two-blocks-per-CU build)
void ht_launch_reset(const ht_model_dev &M, const ht_physics_dev &ph, float *state, const float4 *pts, const int *npts, const float *analysis, const float *cams, const int *list, const int *nlist,
                     int n_unibody, const ht_params &par, float *rows, int *nrows, float *scratch, int scratch_stride, int batch, int B, hipStream_t s, bool many_frames, int n_cu, bool exact)
{
	const size_t cloud = (((size_t)M.plane_off[M.nb] + 16 + 3) & ~(size_t)3) * sizeof(float4) + ((sizeof(closest_lds) + 15) & ~(size_t)15) + HT_MAXNB * BT * sizeof(float);
	const size_t solve = (size_t)(UB_LDS_ROWS + QUAD_CHAIN_SLACK) * (CREC * sizeof(float) + sizeof(float) + sizeof(unsigned short));
	const size_t dyn = cloud > solve ? cloud : solve;
	static size_t attr_set[64];               // per device: the attribute belongs to the device's copy of the code object
	static std::mutex attr_lock;              // contexts of several host threads may launch at the same time: the limit is raised before anybody launches with it
	int dev = 0; (void)hipGetDevice(&dev); dev &= 63;
	std::unique_lock<std::mutex> lk(attr_lock);
	if (attr_set[dev] < dyn)
	{
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_reset<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_reset<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_reset<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
		attr_set[dev] = dyn;
	}
	lk.unlock();
	const bool two = !list || many_frames;
	const int grid = B < (two ? 2 : 1) * n_cu ? B : (two ? 2 : 1) * n_cu;
	if (exact)      // tests only (ht_debug_solver_build 5)
		hipLaunchKernelGGL((k_reset<2, true>), dim3(grid), dim3(RS_THREADS), dyn, s, M, ph, state, pts, npts, analysis, cams, list, nlist, B, n_unibody, par.unibody_force, rows, nrows, scratch, scratch_stride, batch,
		                   ht_tuning_flags());
	else if (!two)
		hipLaunchKernelGGL(k_reset<1>, dim3(grid), dim3(RS_THREADS), dyn, s, M, ph, state, pts, npts, analysis, cams, list, nlist, B, n_unibody, par.unibody_force, rows, nrows, scratch, scratch_stride, batch,
		                   ht_tuning_flags());
	else
		hipLaunchKernelGGL(k_reset<2>, dim3(grid), dim3(RS_THREADS), dyn, s, M, ph, state, pts, npts, analysis, cams, list, nlist, B, n_unibody, par.unibody_force, rows, nrows, scratch, scratch_stride, batch,
		                   ht_tuning_flags());
}
void ht_launch_fit_error(const ht_model_dev &M, const float *state, const float4 *pts, const int *npts, const uint16_t *depth, const float *cams, int w, int h, float scale, float *err, int B, hipStream_t s, const ht_fit_after *after)
{
	ht_fit_after none; memset(&none, 0, sizeof none);
#ifdef HT_TUNING
	{ static int done = 0; if (!done) { const int f = ht_tuning_flags(); (void)hipMemcpyToSymbol(HIP_SYMBOL(ht_fit_error_dbg), &f, sizeof(int)); done = 1; } }
#endif
	hipLaunchKernelGGL(k_fit_error, dim3(B), dim3(256), ((size_t)M.plane_off[M.nb] + 16) * sizeof(float4), s, M, state, pts, npts, depth, cams, w, h, scale, err, after ? *after : none);
}
void ht_launch_chamber(const ht_model_dev &M, const float *state, const float4 *pts, const int *npts, int min_point_num, int enabled, float maxforce, float *rows, int *nch, int B, hipStream_t s,
                       const float *planes, const int *planes_on)
{
	hipLaunchKernelGGL(k_chamber, dim3(B), dim3(CHB_THREADS), 0, s, M, state, pts, npts, min_point_num, enabled, maxforce, rows, nch, planes, planes ? planes_on : nullptr);
}

// order[.] = the frames by their point counts, most first (equal counts in whatever order the atomics fall: nothing depends on it): a counting sort in one block
__global__ __launch_bounds__(1024) void k_order_by_points(const int *__restrict__ npts, int *__restrict__ order, int B)
{
	__shared__ int hist[4096 + 1];
	__shared__ int part[1024];
	const int t = threadIdx.x;
	for (int i = t; i <= 4096; i += 1024) hist[i] = 0;
	__syncthreads();
	for (int b = t; b < B; b += 1024) { int k = npts[b]; k = 4095 - (k < 0 ? 0 : k > 4095 ? 4095 : k); atomicAdd(&hist[k], 1); }
	__syncthreads();
	int loc[4], sum = 0;
	for (int j = 0; j < 4; j++) { loc[j] = sum; sum += hist[4 * t + j]; }
	part[t] = sum;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) { const int v = t >= o ? part[t - o] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
	const int base = part[t] - sum;
	__syncthreads();
	for (int j = 0; j < 4; j++) hist[4 * t + j] = base + loc[j];
	__syncthreads();
	for (int b = t; b < B; b += 1024) { int k = npts[b]; k = 4095 - (k < 0 ? 0 : k > 4095 ? 4095 : k); order[atomicAdd(&hist[k], 1)] = b; }
}
void ht_launch_order_by_points(const int *npts, int *order, int B, hipStream_t s) { hipLaunchKernelGGL(k_order_by_points, dim3(1), dim3(1024), 0, s, npts, order, B); }
