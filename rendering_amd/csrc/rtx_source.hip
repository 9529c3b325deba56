// Source records of the prune blocks (rtxd::PruneRec, DESIGN_HISTORY.md 3.1d): for every wide-node slot the largest P_S of the triangles
// below it, for one source S -- the camera (rebuilt with the view) or a point light.  Included by rtx_api.hip.
//
// What P_S bounds.  The reference's ray / triangle test (objects.cpp:59-95: pvec = dir x e2, det = e1 . pvec, u = tvec . pvec / det,
// qvec = tvec x e1, v = dir . qvec / det, t = e2 . qvec / det; fp32, no FMA) accepts a pair when det_c >= 1e-8, 0 <= u_c <= 1,
// 0 <= v_c, u_c + v_c <= 1, 0 <= t_c (< the ray's limit: scene.cpp:740).  With the exact numerators det = dir . m (m = e2 x e1),
// Nu, Nv, Nt of the SAME fp32 inputs and the rounding errors of the reference's evaluation (DESIGN_HISTORY.md 3.3; u = 2^-24, s1 = |e1|_1,
// s2 = |e2|_1, l1 = |e1|_2, l2 = |e2|_2, ainf = |orig - v0|_inf, dmax = |dir|_inf)
//     |det_c - det| <= ed = 5.1 u dmax s1 s2,  |Nu_c - Nu| <= eu = 12.2 u dmax ainf s2,  |Nv_c - Nv| <= ev = 12.2 u dmax ainf s1,
//     |Nt_c - Nt| <= et = 6.1 u ainf s1 s2
// an accepted pair has
//  (1) det >= 1e-8 - ed > 0                                                        (s1 s2 <= 0.0159, dmax <= 1.001);
//  (2) the EXACT intersection of the ray's line with the triangle's plane, X* = orig + t* dir = v0 + u* e1 + v* e2, at
//      u* >= -delta, v* >= -delta, u* + v* <= 1 + delta,  delta = 6 u + 1.001 (eu + ev + ed) / det   (u_c + v_c <= 1 in fp32, two rounded
//      products), a triangle whose corners lie within  Lambda = delta lsum,  lsum = l1 + l2 + max(l1, l2),  of the triangle's own
//      (the corner (1 + 2 delta, -delta) is delta (2 e1 - e2) from v0 + e1): X* is within Lambda of a point of the triangle;
//  (3) t* in [-et / det, limit (1 + 3 u) (1 + ed / det) + et / det].
// The unconditional bound (rtxd::PruneRec, Pgen) puts det = 1e-8 into (2).  Here every ray of the walk passes through a point S'
// within sigma of S:  S' - X* = (ts - t*) dir and |(S' - X*) . m| = H' |m|_2 with H' the distance of S' from the plane, hence
//     det = H' |dir|_2 |m|_2 / |S' - X*|_2  >=  H mm dmin / (D + Lambda)        (H = H_S - sigma, mm = |m|_2, D = max |S - vertex|_2 + sigma)
// -- the ray cannot graze the triangle's plane more closely than the source's height over it allows.  With eu + ev + ed <= E u dmax
// ainf_rt, where ainf_rt is the value pruneAlive uses (its distance runs to the far side of a box that contains the triangle, so
// ainf_rt >= ainf and ainf_rt >= max(s1, s2) / 6) and
//     E = 12.2 (s1 + s2) + 5.1 s1 s2 / A_T    camera: orig = S exactly, ainf_rt >= A_T = max |S - vertex|_inf
//     E = 27.5 (s1 + s2)                      point light: origins anywhere (s1 s2 <= 3 ainf_rt (s1 + s2)),
// (2) becomes  Lambda <= 6 u lsum + kappa (D + Lambda),  kappa = 1.001 E u dmax ainf_rt lsum / (H mm dmin);  with kappa <= 1/2:
//     Lambda <= 12 u lsum + 2 kappa D.
// The clamped parameter t~ = min(max(t*, 0), limit (1 + 2^-18)) moves the point by at most 0.56 Lambda along the ray (et |dir|_2 / det
// <= 0.25 x, sqrt(3) ainf ed / det <= 0.21 x the kappa term; Lambda ed / det <= omega Lambda with omega <= 0.1 below; the overshoot
// beyond the limit is <= t* ed / det and t* |dir|_2 <= sqrt(3) ainf + Lambda), so orig + t~ dir lies within 1.6 Lambda of the triangle, t~ in
// the segment pruneAlive tests.  pruneAlive inflates by 216 dmax ainf_rt P (216 = 36 u / 1e-8 rounded up) with dmax >= |dir|_2 /
// sqrt(3) >= 0.57, so
//     P_S = P_add + 0.0152 E u lsum D / (H mm)           [2 x 1.6 x 1.001 / (216 x 0.99) = 0.01498]
//     P_add = 3 u                       point light: 216 x 0.57 ainf_rt >= 6.8 lsum (lsum <= 18 ainf_rt) against 19.2 u lsum
//     P_add = 20 u lsum / (123 A_T)     camera
// suffices.  Side conditions, checked per triangle here for the worst admissible ray (ainf_rt <= A = A_T for the camera,
// kSrcAinfMax for a light -- pruneAlive checks it per record -- and |dir|_2 in [0.99, 1.001], checked per walk: traceWave):
// H > 0;  s1 s2 <= 0.0159;  kappa_max <= 1/2;  kappa_max D <= 0.4 (so Lambda <= 1);  omega = 5.1 u 1.001 s1 s2 (D + 1) / (0.99 H mm)
// <= 0.1.  A triangle that fails one keeps Pgen = s1 s2.  tools/research/src_bound_check.py and tests/test_prune_bound_cpu.py run the
// reference's arithmetic on adversarial pairs against this function (through rtx_source_p_probe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "rtx_device.h"

namespace rtxsrc {

// P_S of one triangle (v0, e1, e2 as the exact test sees them: rtxd::RefA / RefB / RefC) for the source S, sigma; Pgen where the
// certificate does not hold.  cam: the rays START at S (the camera).  Double precision throughout: its own rounding (1e-16
// relative) disappears in the factors above.
__host__ __device__ inline float sourceP(const float v0[3], const float e1[3], const float e2[3], const double S[3], double sigma, bool cam)
{
	const double u = 0x1p-24;
	const double s1 = fabs((double)e1[0]) + fabs((double)e1[1]) + fabs((double)e1[2]), s2 = fabs((double)e2[0]) + fabs((double)e2[1]) + fabs((double)e2[2]);
	const double pgen = s1 * s2;
	const float pgenF = (float)(pgen * (1.0 + 0x1p-20) + 1e-37);      // (as flattenMesh rounds a slot's Pgen)
	if (!(pgen > 0.0) || !(pgen <= 0.0159)) return pgenF;             // (zero edge: never accepted -- any P will do; NaN / large: no certificate)
	const double m[3] = { (double)e2[1] * e1[2] - (double)e2[2] * e1[1], (double)e2[2] * e1[0] - (double)e2[0] * e1[2], (double)e2[0] * e1[1] - (double)e2[1] * e1[0] };
	const double mm = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
	if (!(mm > 0.0)) return pgenF;
	const double a[3] = { S[0] - v0[0], S[1] - v0[1], S[2] - v0[2] };
	const double H = fabs(a[0] * m[0] + a[1] * m[1] + a[2] * m[2]) / mm * (1.0 - 1e-9) - sigma;
	if (!(H > 0.0)) return pgenF;
	double D = 0, AT = 0;
	for (int k = 0; k < 3; k++) {
		const double b[3] = { a[0] - (k == 1 ? e1[0] : (k == 2 ? e2[0] : 0.0f)), a[1] - (k == 1 ? e1[1] : (k == 2 ? e2[1] : 0.0f)), a[2] - (k == 1 ? e1[2] : (k == 2 ? e2[2] : 0.0f)) };
		const double l = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
		D = l > D ? l : D;
		for (int c = 0; c < 3; c++) AT = fabs(b[c]) > AT ? fabs(b[c]) : AT;
	}
	D = D * (1.0 + 1e-9) + sigma;
	if (cam && !(AT > 0.0)) return pgenF;
	const double l1 = sqrt((double)e1[0] * e1[0] + (double)e1[1] * e1[1] + (double)e1[2] * e1[2]), l2 = sqrt((double)e2[0] * e2[0] + (double)e2[1] * e2[1] + (double)e2[2] * e2[2]);
	const double lsum = (l1 + l2 + (l1 > l2 ? l1 : l2)) * (1.0 + 1e-9);
	const double ATlo = AT * (1.0 - 1e-6);
	const double E = cam ? 12.2 * (s1 + s2) + 5.1 * pgen / ATlo : 27.5 * (s1 + s2);
	const double Amax = cam ? AT * (1.0 + 1e-6) : (double)rtxd::kSrcAinfMax;
	const double Hmm = H * mm * 0.99;
	const double kappaMax = 1.001 * E * u * 1.001 * Amax * lsum / Hmm;
	const double omega = 5.1 * u * 1.001 * pgen * (D + 1.0) / Hmm;
	if (!(kappaMax <= 0.5) || !(kappaMax * D <= 0.4) || !(omega <= 0.1)) return pgenF;
	const double padd = cam ? 20.0 * u * lsum / (123.0 * ATlo) : 3.0 * u;
	const double ps = padd + 0.0152 * E * u * lsum * D / (H * mm);
	const float psF = (float)(ps * (1.0 + 0x1p-20) + 1e-37);
	return psF < pgenF ? psF : pgenF;
}

// P_S of every leaf reference (refP[r]) and the largest of every 64 consecutive ones (blockP[r / 64]): one thread per reference.
__global__ void __launch_bounds__(256) rtxSourceRefKernel(const rtxd::RefA* __restrict__ refA, const rtxd::RefB* __restrict__ refB, const rtxd::RefC* __restrict__ refC,
                                                          uint32_t nRefs, double Sx, double Sy, double Sz, double sigma, int cam, float* __restrict__ refP, float* __restrict__ blockP)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	float p = 0.0f;
	if (r < nRefs) {
		const rtxd::RefA a = refA[r]; const rtxd::RefB b = refB[r]; const rtxd::RefC c = refC[r];
		const float v0[3] = { a.v0x, a.v0y, a.v0z }, e1[3] = { b.e1x, b.e1y, b.e1z }, e2[3] = { b.e2x, c.e2y, c.e2z };
		const double S[3] = { Sx, Sy, Sz };
		p = sourceP(v0, e1, e2, S, sigma, cam != 0);
		if (!(p == p)) p = __builtin_inff();      // (NaN inputs: no bound -- the largest value wins every max below)
		refP[r] = p;
	}
	float mx = p;
	for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
	if ((threadIdx.x & 63u) == 0 && r < nRefs) blockP[r >> 6] = mx;
}

// One wave per wide-node slot: P of the slot in source copy `copy` = the largest P_S over the references of the slot's subtree
// (slotRange: [begin, end) per slot, a range that COVERS them), never above the slot's Pgen.  Slots without a usable record
// (empty: h < 0; not finite: P = inf) are left as they were copied from copy 0.
__global__ void __launch_bounds__(256) rtxSourceSlotKernel(const uint32_t* __restrict__ slotRange, uint32_t nWide, const float* __restrict__ refP, const float* __restrict__ blockP,
                                                           rtxd::PruneBlock* __restrict__ copy)
{
	const uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (slot >= nWide * rtxd::kWideSlots) return;
	const uint32_t b = slotRange[2 * slot], e = slotRange[2 * slot + 1];
	if (b >= e) return;
	float mx = 0.0f;
	const uint32_t bb = (b + 63) >> 6, be = e >> 6;      // whole blocks of 64 references: [bb, be)
	if (bb >= be) { for (uint32_t r = b + lane; r < e; r += 64) mx = fmaxf(mx, refP[r]); }
	else {
		for (uint32_t r = b + lane; r < (bb << 6); r += 64) mx = fmaxf(mx, refP[r]);
		for (uint32_t j = bb + lane; j < be; j += 64) mx = fmaxf(mx, blockP[j]);
		for (uint32_t r = (be << 6) + lane; r < e; r += 64) mx = fmaxf(mx, refP[r]);
	}
	for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
	if (lane == 0) {
		rtxd::PruneRec& pr = copy[slot / rtxd::kWideSlots].box[slot % rtxd::kWideSlots];
		const float pg = pr.Pgen;
		if (pr.h[0] >= 0.0f && pg < __builtin_inff()) pr.P = mx < pg ? mx : pg;
	}
}

} // namespace rtxsrc
