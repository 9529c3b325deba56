// Microbenchmark: does a wave64 VALU instruction issue faster on gfx950 when only part of EXEC is set?  8 waves per SIMD, 8 independent
// registers per wave; the loop body runs under `if (lane predicate)`, so EXEC inside the loop is the predicate's mask.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/exec_half.bin tools/ubench/exec_half.hip && tools/ubench/exec_half.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(OPSTR) asm volatile(OPSTR " %0, %0, %8, %8\n\t" OPSTR " %1, %1, %8, %8\n\t" OPSTR " %2, %2, %8, %8\n\t" OPSTR " %3, %3, %8, %8\n\t" OPSTR " %4, %4, %8, %8\n\t" OPSTR " %5, %5, %8, %8\n\t" OPSTR " %6, %6, %8, %8\n\t" OPSTR " %7, %7, %8, %8" \
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters)
{
	float x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + a + i;
	float y = b;
	const unsigned lane = threadIdx.x & 63u;
	const bool on = MODE == 0 ? true : (MODE == 1 ? lane < 32 : (MODE == 2 ? lane >= 32 : (MODE == 3 ? lane < 16 : (MODE == 4 ? (lane & 1) == 0 : (MODE == 5 ? (lane & 16) == 0 : lane == 0)))));
	if (on)
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int u = 0; u < 8; u++) { BODY("v_fma_f32") }
		}
	float s = 0; for (int i = 0; i < 8; i++) s += x[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + y;
}
template <int MODE> void run(const char* name, float* out)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4000, blocks = 256 * 8;      // 8 blocks of 4 waves per CU = 8 waves per SIMD
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f, 10);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	// per SIMD: 8 waves x iters x 64 instructions
	const double instr = 8.0 * iters * 64.0;
	printf("%-28s %8.3f ms  %.2f cycles per wave64 v_fma_f32 per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
}
int main()
{
	float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
	run<0>("EXEC = all 64 lanes", out); run<1>("EXEC = lanes 0-31", out); run<2>("EXEC = lanes 32-63", out); run<3>("EXEC = lanes 0-15", out);
	run<4>("EXEC = even lanes", out); run<5>("EXEC = lanes with bit 4 clear", out); run<6>("EXEC = lane 0", out);
	return 0;
}
