// Diagnostic: the v_fma_mix split of csrc/split_f16.h against (half)(v - (float)hi), bit for bit, on 2^21 values over 40
// binades plus edge cases.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/check_split.hip -o check_split && ./check_split
#include <hip/hip_runtime.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned split_lo2(float v0, float v1, unsigned hi) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(v0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(hi));
    return lo;
}
__global__ void k(const float* x, unsigned* hi, unsigned* lo, unsigned* lo_ref) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    float2v v = {x[2*i], x[2*i+1]};
    half2v h = __builtin_convertvector(v, half2v);
    unsigned hu = __builtin_bit_cast(unsigned, h);
    hi[i] = hu; lo[i] = split_lo2(v[0], v[1], hu);
    half2v l = {(_Float16)(v[0] - (float)h[0]), (_Float16)(v[1] - (float)h[1])};
    lo_ref[i] = __builtin_bit_cast(unsigned, l);
}
int main() {
    const int n = 1 << 20;
    float* x; unsigned *hi, *lo, *lr;
    hipMallocManaged(&x, 2*n*4); hipMallocManaged(&hi, n*4); hipMallocManaged(&lo, n*4); hipMallocManaged(&lr, n*4);
    unsigned s = 12345;
    for (int i = 0; i < 2*n; ++i) { s = s*1664525u + 1013904223u; union {unsigned u; float f;} c; c.u = (s & 0x807fffffu) | (((s>>23)%40 + 100) << 23); x[i] = c.f; }
    x[0] = 0.f; x[1] = 65504.f; x[2] = 1e-8f; x[3] = -3.3e-5f;
    hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, x, hi, lo, lr);
    hipDeviceSynchronize();
    int bad = 0; for (int i = 0; i < n; ++i) bad += lo[i] != lr[i];
    printf("mismatches: %d of %d\n", bad, n);
    return bad != 0;
}
