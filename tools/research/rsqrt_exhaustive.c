// Exhaustive proof-by-enumeration for the fast path of invLenD (rendering_amd/csrc/rtx_kernels.hip):
//     reference (geometry.h:104-112):  f = (float)(1.0 / sqrt((double)l2))
//     fast path: y0 = v_rsq_f32(l2) (any value within +-3 ulp of the truth is tried here), one Newton step with exact fp32 residuals (FMA),
//     a second exact residual decides whether the candidate is certainly the reference's value; otherwise the caller takes the fp64 path.
// For every float mantissa and both exponent parities (1/sqrt(m 4^k) = 2^-k / sqrt(m): the fast path's arithmetic is scale-invariant inside the range the
// device function admits, [2^-60, 2^60]), and for every starting value y0 in {RN(truth) + j ulp, j = -3 .. 3}: accepted => bit-identical to the reference.
// Also reports how often the fast path gives up.   gcc -O2 -mfma -o /tmp/rsq tools/research/rsqrt_exhaustive.c -lm && /tmp/rsq
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// the device function's fast path, operation for operation; returns 1 when it accepts (*out = result)
static inline int fast(float l2, float y0, float* out)
{
	const float t = l2 * y0, te = fmaf(l2, y0, -t);
	float e = fmaf(-t, y0, 1.0f); e = fmaf(-te, y0, e);
	const float y1 = fmaf(y0 * 0.5f, e, y0);
	const float t1 = l2 * y1, t1e = fmaf(l2, y1, -t1);
	float e1 = fmaf(-t1, y1, 1.0f); e1 = fmaf(-t1e, y1, e1);
	const float ulp = asf((asu(y1) & 0x7f800000u) - (23u << 23));
	*out = y1;
	return fabsf(e1 * y1) < ulp * 0.99998474f;      // (1 - 2^-16)
}

int main(void)
{
	unsigned long long tried = 0, accepted = 0, wrong = 0, rejected_all = 0;
	const int exps[] = { 127, 128, 127 - 60, 128 + 58, 127 + 33, 127 - 17 };      // both parities near 1, the ends of the admitted range, two in between
	for (int ei = 0; ei < 6; ei++)
		for (uint32_t m = 0; m < (1u << 23); m++) {
			const float x = asf(((uint32_t)exps[ei] << 23) | m);
			const float ref = (float)(1.0 / sqrt((double)x));
			const float c = (float)(1.0 / sqrt((double)x));      // centre of the starting values
			int any = 0;
			for (int j = -3; j <= 3; j++) {
				const float y0 = asf(asu(c) + (uint32_t)j);
				float got;
				tried++;
				if (fast(x, y0, &got)) {
					accepted++; any = 1;
					if (asu(got) != asu(ref)) { if (wrong < 10) printf("WRONG x=%a y0=%a got=%a ref=%a\n", x, y0, got, ref); wrong++; }
				}
			}
			if (!any) rejected_all++;
		}
	printf("tried %llu, accepted %llu (%.5f %%), wrong %llu, inputs rejected for every starting value %llu\n", tried, accepted, 100.0 * accepted / tried, wrong, rejected_all);
	return wrong != 0;
}
