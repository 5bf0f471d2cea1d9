// ht_track.hip -- tracker-level kernels: state plumbing and the user-space pose output.
//
// Reference computations:
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

// (PoseFromScratch and UnibodyFit, the full-reset branch, are one kernel per flagged frame: k_reset, csrc/ht_cloud.hip)

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
void ht_launch_output(const ht_model_dev &M, const float *hand, const int *npts, int *initializing, int min_point_num, float *poses, int n, hipStream_t s, int raw)
{
	hipLaunchKernelGGL(k_output, dim3((n * M.nb + 255) / 256), dim3(256), 0, s, M, hand, npts, initializing, min_point_num, poses, n, raw);
}
