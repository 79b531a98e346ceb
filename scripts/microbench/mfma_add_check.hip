// Is v_mfma_f32_4x4x1f32 with A = 1.0 the same function as v_add_f32?  D[0][j] = C[0][j] + A[0] * B[j]: with A = 1 the
// product is exact, so the result should be round-to-nearest-even(C + B) in every lane.  Compares bit patterns over
// random operands of every class (normal, denormal, zero, inf, NaN, huge cancellation) and times a dependent chain of
// each kind.     hipcc -O3 --offload-arch=gfx950 -o mfma_add_check mfma_add_check.hip && ./mfma_add_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void check_kernel(const uint32_t *a_bits, const uint32_t *b_bits, uint32_t *mismatch, uint32_t *first, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float a = __uint_as_float(a_bits[i < n ? i : 0]), b = __uint_as_float(b_bits[i < n ? i : 0]);
    float4v c = {a, a, a, a};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, b, c, 0, 0, 0);
    float s;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(a), "v"(b));
    const uint32_t m = __float_as_uint(c[0]), v = __float_as_uint(s);
    const bool both_nan = (m & 0x7FFFFFFFu) > 0x7F800000u && (v & 0x7FFFFFFFu) > 0x7F800000u;
    if (i < n && m != v && !both_nan) {
        if (atomicAdd(mismatch, 1u) == 0u) {
            first[0] = a_bits[i];
            first[1] = b_bits[i];
            first[2] = m;
            first[3] = v;
        }
    }
}

template <bool kMfma>
__global__ void chain_kernel(const float *terms, float *out, int n_terms, int reps) {
    float acc = 0.0f;
    float4v c = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < n_terms; k += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(terms + k);
            if (kMfma) {
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, t.x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, t.y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, t.z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, t.w, c, 0, 0, 0);
            } else {
                acc += t.x;
                acc += t.y;
                acc += t.z;
                acc += t.w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = kMfma ? c[0] : acc;
}

static uint32_t rnd(uint64_t &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 32);
}

int main() {
    const int n = 1 << 24;
    uint32_t *ha = (uint32_t *)malloc(4 * n), *hb = (uint32_t *)malloc(4 * n);
    uint64_t s = 12345;
    for (int i = 0; i < n; ++i) {
        uint32_t a = rnd(s), b = rnd(s);
        const uint32_t kind = rnd(s) % 8;
        if (kind == 0) a &= 0x807FFFFFu;                       // a denormal / zero
        if (kind == 1) b &= 0x807FFFFFu;                       // b denormal / zero
        if (kind == 2) { a &= 0x807FFFFFu; b &= 0x807FFFFFu; }
        if (kind == 3) b = (a ^ 0x80000000u) + (rnd(s) % 5) - 2;  // near cancellation
        if (kind == 4) { a = (a & 0x80000000u) | 0x7F800000u; }  // inf
        if (kind == 5) { b = (b & 0x807FFFFFu) | ((rnd(s) % 30 + ((a >> 23) & 0xFF) - 15) & 0xFF) << 23; }  // exponents close to a's
        ha[i] = a;
        hb[i] = b;
    }
    uint32_t *da, *db, *dm, *df;
    hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dm, 4); hipMalloc(&df, 16);
    hipMemcpy(da, ha, 4 * n, hipMemcpyHostToDevice); hipMemcpy(db, hb, 4 * n, hipMemcpyHostToDevice);
    hipMemset(dm, 0, 4); hipMemset(df, 0, 16);
    hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, da, db, dm, df, n);
    uint32_t mism = 0, first[4];
    hipMemcpy(&mism, dm, 4, hipMemcpyDeviceToHost); hipMemcpy(first, df, 16, hipMemcpyDeviceToHost);
    printf("pairs %d  mismatches %u", n, mism);
    if (mism) printf("  first: a %08x b %08x mfma %08x add %08x", first[0], first[1], first[2], first[3]);
    printf("\n");
    // dependent-chain rate: one wave per SIMD pair alone, then 4 waves per SIMD of the VALU kind beside it
    const int n_terms = 448, reps = 2000;
    float *dt, *dout;
    hipMalloc(&dt, 4 * n_terms); hipMalloc(&dout, 4 * 256 * 2048);
    hipMemset(dt, 0, 4 * n_terms);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        for (int kind = 0; kind < 2; ++kind) {
            const dim3 grid(256 * waves_per_simd), block(256);  // 4 waves per block = one per SIMD
            hipEventRecord(e0);
            if (kind) hipLaunchKernelGGL(chain_kernel<true>, grid, block, 0, 0, dt, dout, n_terms, reps);
            else hipLaunchKernelGGL(chain_kernel<false>, grid, block, 0, 0, dt, dout, n_terms, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s chain, %d wave(s) per SIMD: %.2f ns per dependent add\n", kind ? "mfma 4x4x1" : "v_add_f32 ", waves_per_simd, ms * 1e6 / ((double)n_terms * reps));
        }
    }
    return mism ? 1 : 0;
}
