// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the ray kernels are made
// of, on gfx950.  8 waves per SIMD, 8 independent registers per wave, inline asm so that nothing is folded away.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, ASM)                                                                                            \
__global__ void __launch_bounds__(256) NAME(float* out, float a, float b, int iters)                                  \
{                                                                                                                    \
	float x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + a + i;                                       \
	float y = b; unsigned long long m = 0;                                                                           \
	for (int i = 0; i < iters; i++) {                                                                                \
		_Pragma("unroll") for (int u = 0; u < 8; u++) { ASM }                                                        \
	}                                                                                                                \
	float s = 0; for (int i = 0; i < 8; i++) s += x[i];                                                              \
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + y + (float)(unsigned)m;                                         \
}
#define A1(OPSTR) asm volatile(OPSTR " %0, %0, %8\n\t" OPSTR " %1, %1, %8\n\t" OPSTR " %2, %2, %8\n\t" OPSTR " %3, %3, %8\n\t" OPSTR " %4, %4, %8\n\t" OPSTR " %5, %5, %8\n\t" OPSTR " %6, %6, %8\n\t" OPSTR " %7, %7, %8" \
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
KERNEL(k_mul, A1("v_mul_f32"))
KERNEL(k_add, A1("v_add_f32"))
KERNEL(k_sub, A1("v_sub_f32"))
KERNEL(k_min, A1("v_min_f32"))
KERNEL(k_max, A1("v_max_f32"))
KERNEL(k_and, A1("v_and_b32"))
KERNEL(k_addu, A1("v_add_u32"))
KERNEL(k_lshl, A1("v_lshlrev_b32"))
#define A2(OPSTR) asm volatile(OPSTR " %0, %0, %8, %8\n\t" OPSTR " %1, %1, %8, %8\n\t" OPSTR " %2, %2, %8, %8\n\t" OPSTR " %3, %3, %8, %8\n\t" OPSTR " %4, %4, %8, %8\n\t" OPSTR " %5, %5, %8, %8\n\t" OPSTR " %6, %6, %8, %8\n\t" OPSTR " %7, %7, %8, %8" \
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));
KERNEL(k_fma, A2("v_fma_f32"))
KERNEL(k_max3, A2("v_max3_f32"))
KERNEL(k_min3, A2("v_min3_f32"))
KERNEL(k_med3, A2("v_med3_f32"))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));)
KERNEL(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "vcc");)
KERNEL(k_cmp_s, asm volatile("v_cmp_lt_f32 %9, %0, %8\n\tv_cmp_lt_f32 %9, %1, %8\n\tv_cmp_lt_f32 %9, %2, %8\n\tv_cmp_lt_f32 %9, %3, %8\n\tv_cmp_lt_f32 %9, %4, %8\n\tv_cmp_lt_f32 %9, %5, %8\n\tv_cmp_lt_f32 %9, %6, %8\n\tv_cmp_lt_f32 %9, %7, %8"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "s"(m));)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_readlane_b32 s22, %2, 3\n\tv_readlane_b32 s23, %3, 3\n\tv_readlane_b32 s20, %4, 3\n\tv_readlane_b32 s21, %5, 3\n\tv_readlane_b32 s22, %6, 3\n\tv_readlane_b32 s23, %7, 3"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y) : "s20", "s21", "s22", "s23");)
KERNEL(k_dpp, asm volatile("v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y));)
KERNEL(k_mul_s, asm volatile("v_mul_f32 %0, %8, %0\n\tv_mul_f32 %1, %8, %1\n\tv_mul_f32 %2, %8, %2\n\tv_mul_f32 %3, %8, %3\n\tv_mul_f32 %4, %8, %4\n\tv_mul_f32 %5, %8, %5\n\tv_mul_f32 %6, %8, %6\n\tv_mul_f32 %7, %8, %7"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(b));)
#define AS(OPSTR) asm volatile(OPSTR " %0, %8, %0\n\t" OPSTR " %1, %8, %1\n\t" OPSTR " %2, %8, %2\n\t" OPSTR " %3, %8, %3\n\t" OPSTR " %4, %8, %4\n\t" OPSTR " %5, %8, %5\n\t" OPSTR " %6, %8, %6\n\t" OPSTR " %7, %8, %7" \
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(b));
KERNEL(k_sub_s, AS("v_sub_f32"))
KERNEL(k_add_s, AS("v_add_f32"))
KERNEL(k_max_s, AS("v_max_f32"))
KERNEL(k_mul_c, asm volatile("v_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 0.5, %1\n\tv_mul_f32 %2, 0.5, %2\n\tv_mul_f32 %3, 0.5, %3\n\tv_mul_f32 %4, 0.5, %4\n\tv_mul_f32 %5, 0.5, %5\n\tv_mul_f32 %6, 0.5, %6\n\tv_mul_f32 %7, 0.5, %7"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));)
KERNEL(k_fmac, A1("v_fmac_f32"))
KERNEL(k_fma_s, asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(b), "v"(y));)
KERNEL(k_fma2, asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(a));)
KERNEL(k_cnd_s, asm volatile("v_cndmask_b32 %0, %0, %8, %9\n\tv_cndmask_b32 %1, %1, %8, %9\n\tv_cndmask_b32 %2, %2, %8, %9\n\tv_cndmask_b32 %3, %3, %8, %9\n\tv_cndmask_b32 %4, %4, %8, %9\n\tv_cndmask_b32 %5, %5, %8, %9\n\tv_cndmask_b32 %6, %6, %8, %9\n\tv_cndmask_b32 %7, %7, %8, %9"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "s"(m));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
	: "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));)
KERNEL(k_xor, A1("v_xor_b32"))
KERNEL(k_mulu, A1("v_mul_lo_u32"))
#define ADD7 "v_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\tv_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
#define OUT8 "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
KERNEL(k_mix_vcc, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\t" ADD7 : OUT8 : "v"(y) : "vcc");)
KERNEL(k_mix_sgpr, asm volatile("v_cndmask_b32 %0, %0, %8, %9\n\t" ADD7 : OUT8 : "v"(y), "s"(m));)
KERNEL(k_mix_cmp_vcc, asm volatile("v_cmp_lt_f32 vcc, %1, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\t" ADD7 : OUT8 : "v"(y) : "vcc");)
KERNEL(k_mix_cmp_sgpr, asm volatile("v_cmp_lt_f32 s[20:21], %1, %8\n\tv_cndmask_b32 %0, %0, %8, s[20:21]\n\t" ADD7 : OUT8 : "v"(y) : "s20", "s21");)
KERNEL(k_mix_cmp_vcc_far, asm volatile("v_cmp_lt_f32 vcc, %1, %8\n\t" ADD7 "\n\tv_cndmask_b32 %0, %0, %8, vcc" : OUT8 : "v"(y) : "vcc");)
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, ASM)                                                                                           \
__global__ void __launch_bounds__(256) NAME(float* out, float a, float b, int iters)                                  \
{                                                                                                                    \
	f2 x[4]; for (int i = 0; i < 4; i++) x[i] = f2{ threadIdx.x * 1e-3f + a + i, threadIdx.x * 2e-3f + i };          \
	f2 y = { b, a }; f2 sb = { b, b + 1 };                                                                           \
	for (int i = 0; i < iters; i++) {                                                                                \
		_Pragma("unroll") for (int u = 0; u < 8; u++) { ASM }                                                        \
	}                                                                                                                \
	out[blockIdx.x * blockDim.x + threadIdx.x] = x[0].x + x[1].y + x[2].x + x[3].y;                                  \
}
KERNEL2(k_pk_vv, asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(y));)
KERNEL2(k_pk_sv, asm volatile("v_pk_add_f32 %0, %4, %0\n\tv_pk_add_f32 %1, %4, %1\n\tv_pk_add_f32 %2, %4, %2\n\tv_pk_add_f32 %3, %4, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "s"(sb));)
KERNEL2(k_pk_sv_sel, asm volatile("v_pk_add_f32 %0, %4, %0 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %4, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %2, %4, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %4, %3 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "s"(sb));)
KERNEL2(k_pk_mul_sel, asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %3, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(y));)
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3" : "+v"(*(double*)&x[0]), "+v"(*(double*)&x[2]), "+v"(*(double*)&x[4]), "+v"(*(double*)&x[6]));)
template <typename K> void run(const char* name, K kern, int instrPerIter)
{
	float* out; hipMalloc(&out, 256 * 2048 * 4);
	const int iters = 20000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, out, 0.5f, 0.999f, 10);
	hipEventRecord(e0);
	hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, out, 0.5f, 0.999f, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double waveInstr = 2048.0 * 4 * iters * instrPerIter;           // blocks x waves x iterations x instructions
	printf("%-16s %7.3f ms  %5.2f cycles / wave-instruction / SIMD (at 2.4 GHz, 1024 SIMDs)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / waveInstr);
	hipFree(out);
}
int main()
{
	run("v_mul_f32", k_mul, 64); run("v_mul_f32 sgpr", k_mul_s, 64); run("v_add_f32", k_add, 64); run("v_sub_f32", k_sub, 64); run("v_fma_f32", k_fma, 64);
	run("v_min_f32", k_min, 64); run("v_max_f32", k_max, 64); run("v_max3_f32", k_max3, 64); run("v_min3_f32", k_min3, 64); run("v_med3_f32", k_med3, 64);
	run("v_cndmask_b32", k_cndmask, 64); run("v_mov_b32", k_mov, 64); run("v_cmp vcc", k_cmp, 64); run("v_cmp sgpr", k_cmp_s, 64);
	run("v_and_b32", k_and, 64); run("v_add_u32", k_addu, 64); run("v_lshlrev_b32", k_lshl, 64);
	run("v_sub_f32 sgpr", k_sub_s, 64); run("v_add_f32 sgpr", k_add_s, 64); run("v_max_f32 sgpr", k_max_s, 64); run("v_mul_f32 const", k_mul_c, 64);
	run("v_fmac_f32", k_fmac, 64); run("v_fma_f32 v,s,v", k_fma_s, 64); run("v_fma_f32 v,v,v'", k_fma2, 64); run("v_cndmask sgpr", k_cnd_s, 64); run("v_rcp_f32", k_rcp, 64);
	run("v_xor_b32", k_xor, 64); run("v_mul_lo_u32", k_mulu, 64);
	run("v_pk_add v,v", k_pk_vv, 32); run("v_pk_add s,v", k_pk_sv, 32); run("v_pk_add s,v sel/neg", k_pk_sv_sel, 32); run("v_pk_mul v,v sel", k_pk_mul_sel, 32);
	run("cnd vcc + 7 add", k_mix_vcc, 64); run("cnd sgpr + 7 add", k_mix_sgpr, 64); run("cmp,cnd vcc + 7 add", k_mix_cmp_vcc, 72); run("cmp,cnd sgpr+7add", k_mix_cmp_sgpr, 72); run("cmp vcc,7add,cnd", k_mix_cmp_vcc_far, 72);
	run("v_readlane_b32", k_readlane, 64); run("v_max_f32_dpp", k_dpp, 64); run("v_fma_f64", k_fma_f64, 32);
	return 0;
}
