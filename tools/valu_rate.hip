// Development micro-benchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950 as a function
// of waves per SIMD (decides whether 2-cells-per-lane packed math can pay).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = b0 + 1.0f, b5 = b1 + 1.0f, b6 = b2 + 1.0f, b7 = b3 + 1.0f;
    const float c = 1.0000001f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // scalar add, 8 independent chains
            a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
        } else if (MODE == 1) {  // scalar mul
            a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
        } else if (MODE == 2) {  // packed add
            b0 += c; b1 += c; b2 += c; b3 += c; b4 += c; b5 += c; b6 += c; b7 += c;
        } else if (MODE == 3) {  // packed mul
            b0 *= c; b1 *= c; b2 *= c; b3 *= c; b4 *= c; b5 *= c; b6 *= c; b7 *= c;
        } else if (MODE == 4) {  // scalar fma
            a0 = __builtin_fmaf(a0, c, c); a1 = __builtin_fmaf(a1, c, c); a2 = __builtin_fmaf(a2, c, c); a3 = __builtin_fmaf(a3, c, c);
            a4 = __builtin_fmaf(a4, c, c); a5 = __builtin_fmaf(a5, c, c); a6 = __builtin_fmaf(a6, c, c); a7 = __builtin_fmaf(a7, c, c);
        } else {  // cndmask-like select
            a0 = a1 > c ? a0 : a2; a1 = a2 > c ? a1 : a3; a2 = a3 > c ? a2 : a4; a3 = a4 > c ? a3 : a5;
            a4 = a5 > c ? a4 : a6; a5 = a6 > c ? a5 : a7; a6 = a7 > c ? a6 : a0; a7 = a0 > c ? a7 : a1;
        }
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b1.y + b2.x + b3.y + b4.x + b5.y + b6.x + b7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
int run(const char* name, float* out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: block = wps*4 waves, one block per CU
        dim3 block(wps * 4 * 64), grid(256);
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 100, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters, 1.0f);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double inst_per_simd = (double)iters * 8 * wps;
        printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n",
               name, wps, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4));
    run<0>("v_add_f32", out); run<1>("v_mul_f32", out); run<4>("v_fma_f32", out);
    run<2>("v_pk_add_f32", out); run<3>("v_pk_mul_f32", out); run<5>("cmp+cndmask", out);
    return 0;
}
