// Microbenchmark: wave64 issue rate of the integer VALU instructions the tree kernels are made of.
// Each lane runs ITER iterations of 8 independent chains of one instruction kind; waves/SIMD swept.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int KIND>
__global__ void k(int* out, int iters, int seed)
{
    int x[8];
    int a = threadIdx.x * 3 + seed, b = blockIdx.x + 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) x[i] = x[i] + b;                               // v_add_u32
            if (KIND == 1) x[i] = __mul24(x[i], b) ;                       // v_mul_i32_i24
            if (KIND == 2) x[i] = x[i] * b;                                // v_mul_lo_u32
            if (KIND == 3) x[i] = min(max(x[i] + b, -65536), 65535);       // add + med3
            if (KIND == 4) x[i] = ((unsigned)(x[i] + b + 65536) > 131071u) ? 0 : x[i] + b; // add, add, cmp, cndmask
            if (KIND == 5) x[i] = (x[i] >> 3) + b;                         // ashr + add
            if (KIND == 6) { float f = __int_as_float(x[i]); f = __builtin_fmaf(f, 1.0001f, 0.5f); x[i] = __float_as_int(f); } // v_fma_f32
            if (KIND == 7) x[i] = __mul24(x[i], b) + (x[i] >> 8);          // mad24-ish
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int ops_per_elem)
{
    int* d;
    hipMalloc(&d, 256 * 2048 * 4 * 4);
    const int iters = 4096;
    for (int wps = 1; wps <= 8; wps *= 2) {
        dim3 grid(256 * wps), blk(256);  // wps workgroups of 4 waves per CU -> wps waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, grid, blk, 0, 0, d, 64, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, grid, blk, 0, 0, d, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr_per_simd = (double)iters * 8 * ops_per_elem * wps;  // per SIMD
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD=%d  %.3f ms  ~%.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, ms, cyc / wave_instr_per_simd);
    }
    hipFree(d);
}

int main()
{
    run<0>("v_add_u32", 1);
    run<1>("v_mul_i32_i24", 1);
    run<2>("v_mul_lo_u32", 1);
    run<3>("add+med3 (2 ops)", 2);
    run<4>("add,add,cmp,cndmask (4 ops)", 4);
    run<5>("ashr+add (2 ops)", 2);
    run<6>("v_fma_f32", 1);
    run<7>("mul24+ashr+add (3 ops)", 3);
    return 0;
}
