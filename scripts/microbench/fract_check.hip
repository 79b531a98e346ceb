// Experiment (GPU box): is v_fract_f32(x) == x - (float)(int)x bit for bit for every 0 <= x < 2^31 (denormals included)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -o /tmp/fract_check fract_check.hip && /tmp/fract_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void check(unsigned long long *bad, unsigned *first) {
    // every non-negative float below 2^31: bit patterns 0 .. 0x4EFFFFFF
    const unsigned long long n = 0x4F000000ull;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float a = __builtin_amdgcn_fractf(x);
        const float b = x - (float)(int)x;
        if (__float_as_uint(a) != __float_as_uint(b)) {
            if (atomicAdd(bad, 1ull) == 0ull) {
                *first = (unsigned)i;
            }
        }
    }
}
int main() {
    unsigned long long *bad; unsigned *first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long h = 0; unsigned f = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("patterns checked: %llu, mismatches: %llu, first: 0x%08x\n", 0x4F000000ull, h, f);
    return h != 0;
}
