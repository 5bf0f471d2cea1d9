/*
 * ho_cnn.c -- CPU restatement of the reference CNN forward pass (third_party/cnn.h) with the topology of
 * PoseInitializerCNN (include/handtrack.h:103-130).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  Parity pinned: bit-exact against the reference's own CNN::Eval on the
 * golden fixtures (tests/test_oracle_vs_golden.py) when both are built -ffp-contract=off.
 *
 * Accumulation order is what defines the bits:
 *   LConv::forward  cnn.h:205-257  out = bias; then taps in rect_iteration order (kx fastest, then ky), then ic: out += in*w
 *   LFull::forward  cnn.h:405-429  Y = B; for i ascending: Y[j] += x[i]*W[i*N+j]
 *   LMaxPool        cnn.h:141-148  max(max(max(a,b),c),d), std::max semantics
 *   TanH::f         cnn.h:29-33    e = expf(2t); (e-1)/(e+1)   (NaN for large t is intentional)
 *   LSoftMaxChunked cnn.h:497-511  y=expf(x); per chunk: sum ascending, then divide
 */
#include <stdlib.h>
#include <string.h>
#include "ht_oracle.h"

static inline float mx(float a, float b) { return (a < b) ? b : a; }

/* weights in .cnnb layout: conv W index = kx + KW*(ky + KH*(ic + IC*oc))  (make_packed_stride cnn.h:45-47) */
static void conv_valid(const float *in, int iw, int ih, int ic, const float *W, const float *B, int kw, int kh, int oc, float *out)
{
	int ow = iw - kw + 1, oh = ih - kh + 1;
	for (int z = 0; z < oc; z++) for (int i = 0; i < ow * oh; i++) out[z * ow * oh + i] = B[z];
	for (int ky = 0; ky < kh; ky++) for (int kx = 0; kx < kw; kx++)
		for (int iz = 0; iz < ic; iz++) for (int oz = 0; oz < oc; oz++)
		{
			float w = W[kx + kw * (ky + kh * (iz + ic * oz))];
			for (int y = 0; y < oh; y++)
			{
				const float *ip = in + iz * iw * ih + (y + ky) * iw + kx;
				float *op = out + oz * ow * oh + y * ow;
				for (int x = 0; x < ow; x++) op[x] += ip[x] * w;
			}
		}
}
static void tanh_ref(float *x, int n) { for (int i = 0; i < n; i++) { float e = expf(2 * x[i]); x[i] = (e - 1) / (e + 1); } }
static void maxpool2(const float *in, int w, int h, int c, float *out)
{
	int ow = w / 2, oh = h / 2;
	for (int z = 0; z < c; z++) for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++)
	{
		const float *p = in + z * w * h;
		out[z * ow * oh + y * ow + x] = mx(mx(mx(p[(2 * y) * w + 2 * x], p[(2 * y) * w + 2 * x + 1]), p[(2 * y + 1) * w + 2 * x]), p[(2 * y + 1) * w + 2 * x + 1]);
	}
}
static void full(const float *x, int M, const float *W, const float *B, int N, float *Y)
{
	memcpy(Y, B, sizeof(float) * N);
	for (int i = 0; i < M; i++) { float xi = x[i]; const float *w = W + (size_t)i * N; for (int j = 0; j < N; j++) Y[j] += xi * w[j]; }
}

/* layer outputs kept by index like CNN::Eval's outputs[] (cnn.h:550-556); layers==NULL to skip.
 * `side` = 64: the topology of PoseInitializerCNN (handtrack.h:108-118).  `side` = 128: the same layer list on a 128x128 input (SURVEY 8d
 * "config 5 (ii)": conv5 -> 124, pool -> 62 -> 31, conv4 -> 28, pool -> 14, FC 12544 -> 2048 -> 2304), pinned against the reference's own
 * LConv / LMaxPool / LFull / LActivation<TanH> / LSoftMaxChunked classes assembled that way (tests/golden/cnn128.htfx). */
void ho_cnn_eval_sized(const float *weights, int side, const float *input, float *output, float *const *layers)
{
	const int c1 = side - 4, p1 = c1 / 2, p2 = p1 / 2, c2 = p2 - 3, p3 = c2 / 2, feat = 64 * p3 * p3;
	const float *W1 = weights, *B1 = W1 + 400, *W2 = B1 + 16, *B2 = W2 + 16384, *W3 = B2 + 64, *B3 = W3 + (size_t)feat * 2048, *W4 = B3 + 2048, *B4 = W4 + (size_t)2048 * 2304;
	const size_t big = (size_t)16 * c1 * c1;
	float *a = malloc(sizeof(float) * big), *b = malloc(sizeof(float) * big);
#define KEEP(i, p, n) do { if (layers && layers[i]) memcpy(layers[i], p, sizeof(float) * (n)); } while (0)
	conv_valid(input, side, side, 1, W1, B1, 5, 5, 16, a); KEEP(0, a, big);
	tanh_ref(a, (int)big);                                KEEP(1, a, big);
	maxpool2(a, c1, c1, 16, b);                           KEEP(2, b, 16 * p1 * p1);
	maxpool2(b, p1, p1, 16, a);                           KEEP(3, a, 16 * p2 * p2);
	conv_valid(a, p2, p2, 16, W2, B2, 4, 4, 64, b);       KEEP(4, b, 64 * c2 * c2);
	tanh_ref(b, 64 * c2 * c2);                            KEEP(5, b, 64 * c2 * c2);
	maxpool2(b, c2, c2, 64, a);                           KEEP(6, a, feat);
	full(a, feat, W3, B3, 2048, b);                       KEEP(7, b, 2048);
	tanh_ref(b, 2048);                                    KEEP(8, b, 2048);
	full(b, 2048, W4, B4, 2304, a);                       KEEP(9, a, 2304);
	for (int i = 0; i < 2304; i++) a[i] = expf(a[i]);
	for (int c = 0, base = 0; c < 24; c++)
	{
		int s = c < 8 ? 256 : 16;
		float sum = 0.0f;
		for (int i = base; i < base + s; i++) sum += a[i];
		for (int i = base; i < base + s; i++) a[i] /= sum;
		base += s;
	}
	KEEP(10, a, 2304);
	memcpy(output, a, sizeof(float) * 2304);
	free(a); free(b);
#undef KEEP
}
void ho_cnn_eval(const float *weights, const float *input, float *output, float *const *layers) { ho_cnn_eval_sized(weights, 64, input, output, layers); }

/* depth -> CNN input, handtrack.h:700 */
void ho_cnn_input(const uint16_t *depth, int n, float depth_scale, float drange_x, float drange_y, float *out)
{
	for (int i = 0; i < n; i++)
	{
		float v = 1.0f - (depth[i] * depth_scale - drange_x) / (drange_y - drange_x);
		out[i] = ((v < 0.0f ? 0.0f : v) > 1.0f) ? 1.0f : (v < 0.0f ? 0.0f : v);   /* clamp = min(max(a,0),1) */
	}
}

/* ---- training step --------------------------------------------------------------------------------
 * CNN::Train cnn.h:558-580: forward keeping every layer's output, E = y - t (mse = mean E^2), errors back through the layers
 * (backward() of layers 10..1 with the weights as they are), then update() of every layer with step alpha.
 *   LSoftMaxChunked::backward cnn.h:512-526, LFull::backward/update :430-445, LActivation::backward :464-469 (TanH::df = 1 - y*y),
 *   LMaxPool::backward :149-164 (first maximum of the 2x2 block in x-then-y order gets the error), LConv::backward/update :258-279
 *   (madd order: output positions x fastest, then y, then output channel; inside, taps kx fastest, then ky, then input channel). */
static void maxpool2_back(const float *X, int w, int h, int c, const float *E, float *D)
{
	int ow = w / 2, oh = h / 2;
	memset(D, 0, sizeof(float) * w * h * c);
	for (int z = 0; z < c; z++) for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++)
	{
		const float *p = X + z * w * h;
		int mxi = (2 * y) * w + 2 * x;
		for (int vy = 0; vy < 2; vy++) for (int vx = 0; vx < 2; vx++) { int q = (2 * y + vy) * w + 2 * x + vx; if (p[q] > p[mxi]) mxi = q; }
		D[z * w * h + mxi] = E[z * ow * oh + y * ow + x];
	}
}
static void conv_back(const float *E, int iw, int ih, int ic, const float *W, int kw, int kh, int oc, float *D)
{
	int ow = iw - kw + 1, oh = ih - kh + 1;
	memset(D, 0, sizeof(float) * iw * ih * ic);
	for (int oz = 0; oz < oc; oz++) for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++)
	{
		const float e = E[oz * ow * oh + y * ow + x];
		for (int iz = 0; iz < ic; iz++) for (int ky = 0; ky < kh; ky++) for (int kx = 0; kx < kw; kx++)
			D[iz * iw * ih + (y + ky) * iw + x + kx] += W[kx + kw * (ky + kh * (iz + ic * oz))] * e;
	}
}
static void conv_update(const float *X, const float *E, int iw, int ih, int ic, float *W, float *B, int kw, int kh, int oc, float alpha)
{
	int ow = iw - kw + 1, oh = ih - kh + 1;
	for (int oz = 0; oz < oc; oz++) for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++)
	{
		const float e = E[oz * ow * oh + y * ow + x], s = -alpha * e;
		for (int iz = 0; iz < ic; iz++) for (int ky = 0; ky < kh; ky++) for (int kx = 0; kx < kw; kx++)
			W[kx + kw * (ky + kh * (iz + ic * oz))] += X[iz * iw * ih + (y + ky) * iw + x + kx] * s;
		B[oz] -= e * alpha;
	}
}
static void full_back(const float *W, int M, int N, const float *E, float *D)
{
	for (int i = 0; i < M; i++) { float d = 0.0f; const float *w = W + (size_t)i * N; for (int j = 0; j < N; j++) d += w[j] * E[j]; D[i] = d; }
}
static void full_update(float *W, float *B, int M, int N, const float *X, const float *E, float alpha)
{
	for (int j = 0; j < N; j++) B[j] -= E[j] * alpha;
	for (int i = 0; i < M; i++) { float *w = W + (size_t)i * N; for (int j = 0; j < N; j++) w[j] -= X[i] * E[j] * alpha; }
}
static void tanh_back(const float *Y, const float *E, int n, float *D) { for (int i = 0; i < n; i++) D[i] = (1.0f - Y[i] * Y[i]) * E[i]; }

float ho_cnn_train(float *weights, const float *input, const float *target, float alpha)
{
	float *W1 = weights, *B1 = W1 + 400, *W2 = B1 + 16, *B2 = W2 + 16384, *W3 = B2 + 64, *B3 = W3 + (size_t)2304 * 2048, *W4 = B3 + 2048, *B4 = W4 + (size_t)2048 * 2304;
	static const int N[11] = { 57600, 57600, 14400, 3600, 9216, 9216, 2304, 2048, 2048, 2304, 2304 };
	float *out[11], *err[11], y[2304];
	for (int i = 0; i < 11; i++) { out[i] = malloc(sizeof(float) * N[i]); err[i] = malloc(sizeof(float) * N[i]); }
	ho_cnn_eval(weights, input, y, out);
	float mse = 0;
	for (int i = 0; i < 2304; i++) { float e = out[10][i] - target[i]; mse += e * e; err[10][i] = e; }
	mse /= 2304;
	/* errors[i-1] = layers[i]->backward(outputs[i-1], outputs[i], errors[i]) */
	for (int c = 0, base = 0; c < 24; c++)      /* layer 10: chunked softmax */
	{
		int s = c < 8 ? 256 : 16;
		float dp = 0.0f;
		for (int i = base; i < base + s; i++) dp += err[10][i] * out[10][i];
		for (int i = base; i < base + s; i++) err[9][i] = out[10][i] * (err[10][i] - dp);
		base += s;
	}
	full_back(W4, 2048, 2304, err[9], err[8]);                 /* layer 9 */
	tanh_back(out[8], err[8], 2048, err[7]);                   /* layer 8 */
	full_back(W3, 2304, 2048, err[7], err[6]);                 /* layer 7 */
	maxpool2_back(out[5], 12, 12, 64, err[6], err[5]);         /* layer 6 */
	tanh_back(out[5], err[5], 9216, err[4]);                   /* layer 5 */
	conv_back(err[4], 15, 15, 16, W2, 4, 4, 64, err[3]);       /* layer 4 */
	maxpool2_back(out[2], 30, 30, 16, err[3], err[2]);         /* layer 3 */
	maxpool2_back(out[1], 60, 60, 16, err[2], err[1]);         /* layer 2 */
	tanh_back(out[1], err[1], 57600, err[0]);                  /* layer 1 */
	/* updates, layer 0 first */
	conv_update(input, err[0], 64, 64, 1, W1, B1, 5, 5, 16, alpha);
	conv_update(out[3], err[4], 15, 15, 16, W2, B2, 4, 4, 64, alpha);
	full_update(W3, B3, 2304, 2048, out[6], err[7], alpha);
	full_update(W4, B4, 2048, 2304, out[8], err[9], alpha);
	for (int i = 0; i < 11; i++) { free(out[i]); free(err[i]); }
	return mse;
}

/* ---- labels: GatherHandExpectedCNN handtrack.h:160-173 -------------------------------------------------------------------
 * ImageFeaturePoints :92-96, RenderHeatMap / NormalizeHeatMap misc_image.h:246-272, HandPoseToKeyAngleSet handtrack.h:132-151,
 * Render1DHeatMaps misc_image.h:281-295, GrayScaleToFloat misc_image.h:171.  atan2/asin/acos/exp/pow without std:: are the C
 * double functions; std::exp(float) in RenderHeatMap is expf.  hcam = camsub(tile camera, 4). */
static unsigned char to_gray(float x) { float v = x * 255.0f; v = ho_clampf(v, 0.0f, 255.0f); return (unsigned char)v; }
void ho_expected_cnn(const float *pose7, const ho_camera *hcam, float *expected2304, float *vals16)
{
	static const struct { int bone; float off[3]; } FP[8] = { { 1, { 0, 0, 0 } }, { 1, { -0.03f, 0, -0.03f } }, { 1, { 0.03f, 0, -0.03f } }, { 4, { 0, 0, 0 } }, { 7, { 0, 0, 0 } }, { 10, { 0, 0, 0 } }, { 13, { 0, 0, 0 } }, { 16, { 0, 0, 0 } } };
	pose_t P[17];
	for (int b = 0; b < 17; b++) P[b] = POSE(F3(pose7[7 * b], pose7[7 * b + 1], pose7[7 * b + 2]), F4(pose7[7 * b + 3], pose7[7 * b + 4], pose7[7 * b + 5], pose7[7 * b + 6]));
	const pose_t ci = pose_inverse(hcam->pose);
	unsigned char img[2304];
	memset(img, 0, sizeof img);
	for (int k = 0; k < 8; k++)
	{
		f3 v = pose_apply(ci, pose_apply(P[FP[k].bone], F3(FP[k].off[0], FP[k].off[1], FP[k].off[2])));
		f2 peak = { v.x / v.z * hcam->focal.x + hcam->principal.x, v.y / v.z * hcam->focal.y + hcam->principal.y };
		unsigned char *h = img + 256 * k;
		int hx = (int)peak.x, hy = (int)peak.y;
		for (int y = ho_maxi(0, hy - 2); y < ho_mini(hcam->h, hy + 3); y++) for (int x = ho_maxi(0, hx - 2); x < ho_mini(hcam->w, hx + 3); x++)
		{
			float dx = peak.x - (float)x, dy = peak.y - (float)y;
			float d2 = dx * dx + dy * dy;
			h[y * hcam->w + x] = to_gray(expf(-d2 / (2.0f * 0.33f)));
		}
		int sum = 0; for (int i = 0; i < 256; i++) sum += h[i];
		if (sum) for (int i = 0; i < 256; i++) h[i] = (unsigned char)(h[i] * 255 / sum);
	}
	/* key angles */
	float vals[16]; int nv = 0;
	f4 palmq = qmul(pose_inverse(hcam->pose).orientation, P[1].orientation);
	vals[nv++] = (float)(atan2((double)qxdir(palmq).x, (double)-qxdir(palmq).z) / (double)(3.14159f * 2.0f) + (double)0.5f);
	vals[nv++] = (float)(asin((double)ho_clampf(qzdir(palmq).z, -1.0f, 1.0f)) / (double)3.14159f + (double)0.5f);
	vals[nv++] = (float)(asin((double)ho_clampf(qzdir(palmq).x, -1.0f, 1.0f)) / (double)3.14159f + (double)0.5f);
	vals[nv++] = (float)(acos((double)dot3(qxdir(P[1].orientation), qzdir(P[4].orientation))) / (double)3.14159f);
	{ static const int bid[4] = { 6, 9, 12, 15 }; for (int k = 0; k < 4; k++) vals[nv++] = (float)(acos((double)ho_clampf(dot3(qydir(P[1].orientation), qydir(P[bid[k]].orientation)), -1.0f, 1.0f)) / (double)3.14159f); }
	{ f3 pz = qzdir(palmq); vals[nv++] = (float)((double)0.5f + atan2((double)-pz.x, (double)-pz.y) / (double)(3.14159f * 2.0f)); }
	while (nv < 16) vals[nv++] = 0.0f;
	if (vals16) memcpy(vals16, vals, sizeof vals);
	unsigned char *vm = img + 2048;
	for (int y = 0; y < 16; y++)
	{
		float v = vals[y] * (float)(16 - 1);
		int sum = 0;
		for (int x = ho_maxi(0, (int)v - 2); x < ho_mini(16, (int)v + 3); x++)
		{
			float d2 = (float)pow((double)((float)x - v), (double)2.0f);
			sum += vm[y * 16 + x] = to_gray((float)exp((double)(-d2 / (2.0f * 0.5f))));
		}
		for (int x = ho_maxi(0, (int)v - 2); sum && x < ho_mini(16, (int)v + 3); x++) vm[y * 16 + x] = (unsigned char)(vm[y * 16 + x] * 255 / sum);
	}
	for (int i = 0; i < 2304; i++) expected2304[i] = img[i] / 255.0f;
}
