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

/* layer outputs kept by index like CNN::Eval's outputs[] (cnn.h:550-556); layers==NULL to skip */
void ho_cnn_eval(const float *weights, const float *input, float *output, float *const *layers)
{
	const float *W1 = weights, *B1 = W1 + 400, *W2 = B1 + 16, *B2 = W2 + 16384, *W3 = B2 + 64, *B3 = W3 + (size_t)2304 * 2048, *W4 = B3 + 2048, *B4 = W4 + (size_t)2048 * 2304;
	float *a = malloc(sizeof(float) * 57600), *b = malloc(sizeof(float) * 57600);
#define KEEP(i, p, n) do { if (layers && layers[i]) memcpy(layers[i], p, sizeof(float) * (n)); } while (0)
	conv_valid(input, 64, 64, 1, W1, B1, 5, 5, 16, a);   KEEP(0, a, 57600);
	tanh_ref(a, 57600);                                   KEEP(1, a, 57600);
	maxpool2(a, 60, 60, 16, b);                           KEEP(2, b, 14400);
	maxpool2(b, 30, 30, 16, a);                           KEEP(3, a, 3600);
	conv_valid(a, 15, 15, 16, W2, B2, 4, 4, 64, b);       KEEP(4, b, 9216);
	tanh_ref(b, 9216);                                    KEEP(5, b, 9216);
	maxpool2(b, 12, 12, 64, a);                           KEEP(6, a, 2304);
	full(a, 2304, W3, B3, 2048, b);                       KEEP(7, b, 2048);
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

/* depth -> CNN input, handtrack.h:700 */
void ho_cnn_input(const uint16_t *depth, int n, float depth_scale, float drange_x, float drange_y, float *out)
{
	for (int i = 0; i < n; i++)
	{
		float v = 1.0f - (depth[i] * depth_scale - drange_x) / (drange_y - drange_x);
		out[i] = ((v < 0.0f ? 0.0f : v) > 1.0f) ? 1.0f : (v < 0.0f ? 0.0f : v);   /* clamp = min(max(a,0),1) */
	}
}
