// Microbenchmark: issue rate of v_mul_f32 / v_pk_mul_f32 / v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_cmp (SGPR result) /
// v_max3_f32 / v_rcp_f32 on gfx950 (wave64), 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters)
{
	float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	f2 p0 = { x0, x1 }, p1 = { x2, x3 }, p2 = { x4, x5 }, p3 = { x6, x7 };
	const f2 bb = { b, b };
	for (int i = 0; i < iters; i++) {
		if (MODE == 0) {
#pragma unroll
			for (int u = 0; u < 8; u++) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b; }
		}
		else if (MODE == 1) {
#pragma unroll
			for (int u = 0; u < 8; u++) { p0 *= bb; p1 *= bb; p2 *= bb; p3 *= bb; }
		}
		else if (MODE == 3) {
#pragma unroll
			for (int u = 0; u < 8; u++) { p0 = __builtin_elementwise_fma(p0, bb, p1); p1 = __builtin_elementwise_fma(p1, bb, p2); p2 = __builtin_elementwise_fma(p2, bb, p3); p3 = __builtin_elementwise_fma(p3, bb, p0); }
		}
		else if (MODE == 4) {
#pragma unroll
			for (int u = 0; u < 8; u++) { p0 += bb; p1 += bb; p2 += bb; p3 += bb; }
		}
		else if (MODE == 5) {
#pragma unroll
			for (int u = 0; u < 8; u++) { x0 = __builtin_fmaxf(__builtin_fmaxf(x0, x1), b); x1 = __builtin_fmaxf(__builtin_fmaxf(x1, x2), b); x2 = __builtin_fmaxf(__builtin_fmaxf(x2, x3), b); x3 = __builtin_fmaxf(__builtin_fmaxf(x3, x0), a); }
		}
		else if (MODE == 6) {
#pragma unroll
			for (int u = 0; u < 8; u++) { x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3); }
		}
		else if (MODE == 7) {
			// v_cmp writing an SGPR pair + scalar use
			unsigned long long acc = 0;
#pragma unroll
			for (int u = 0; u < 8; u++) { acc += __builtin_amdgcn_ballot_w64(x0 < b + u); acc ^= __builtin_amdgcn_ballot_w64(x1 < a + u); acc += __builtin_amdgcn_ballot_w64(x2 > b + u); acc ^= __builtin_amdgcn_ballot_w64(x3 > a + u); }
			x4 += (float)(unsigned)(acc & 1);
		}
		else {
#pragma unroll
			for (int u = 0; u < 8; u++) { x0 = __builtin_fmaf(x0, b, a); x1 = __builtin_fmaf(x1, b, a); x2 = __builtin_fmaf(x2, b, a); x3 = __builtin_fmaf(x3, b, a);
			                              x4 = __builtin_fmaf(x4, b, a); x5 = __builtin_fmaf(x5, b, a); x6 = __builtin_fmaf(x6, b, a); x7 = __builtin_fmaf(x7, b, a); }
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char* name, int instrPerIter, int flopsPerInstrPerLane)
{
	float* out; hipMalloc(&out, 256 * 2048 * 4);
	const int iters = 20000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, out, 0.5f, 0.999f, 10);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, out, 0.5f, 0.999f, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double waveInstr = 2048.0 * 4 * iters * instrPerIter;            // wave-instructions
	double perSimdCycles = ms * 1e-3 * 2.4e9 / (waveInstr / 1024.0);  // cycles per wave-instruction per SIMD at 2.4 GHz
	printf("%-14s %.3f ms  %.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %.1f TFLOP/s\n", name, ms, perSimdCycles,
	       waveInstr * 64 * flopsPerInstrPerLane / (ms * 1e-3) / 1e12);
	hipFree(out);
}
int main() { run<0>("v_mul_f32", 64, 1); run<1>("v_pk_mul_f32", 32, 2); run<2>("v_fma_f32", 64, 2);
	run<3>("v_pk_fma_f32", 32, 4); run<4>("v_pk_add_f32", 32, 2); run<5>("v_max3_f32", 32, 1); run<6>("v_rcp_f32", 32, 1); run<7>("v_cmp->sgpr", 32, 1);
	return 0; }
