// Does s_barrier wait only for the SURVIVING waves of a workgroup once others have ended (CDNA3 ISA, S_BARRIER: "If some waves in the
// threadgroup have already terminated, this waits on only the surviving waves")?  HIP leaves a barrier behind a partial exit undefined; the
// hardware rule is what a kernel whose waves leave one by one (one consumer wave serving three feature waves) would stand on.
// Four waves; wave w leaves after 100 (w + 1) rounds, the last wave goes on to 1 000 rounds; every round each live wave adds 1 to an LDS
// counter between two barriers and checks that the counter grew by exactly the number of live waves.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/barrier_after_exit scripts/microbench/barrier_after_exit.hip && timeout -k 5 30 /tmp/barrier_after_exit
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) leave_one_by_one(int *out) {
    __shared__ int counter;
    __shared__ int live;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        counter = 0;
        live = 4;
    }
    __syncthreads();
    const int my_rounds = wave == 3 ? 1000 : 100 * (wave + 1);
    int bad = 0;
    for (int round = 0; round < my_rounds; ++round) {
        const int before = counter, alive = live;
        __syncthreads();
        if (lane == 0) {
            atomicAdd(&counter, 1);
        }
        __syncthreads();
        if (counter != before + alive) {
            ++bad;
        }
        if (round == my_rounds - 1 && lane == 0) {
            atomicSub(&live, 1);  // before this round's last barrier: the others read it behind that barrier
        }
        __syncthreads();  // everybody has read the counter before the next round changes it
    }
    if (lane == 0) {
        out[blockIdx.x * 4 + wave] = bad;
    }
    __builtin_amdgcn_s_waitcnt(0);  // the LDS update is done before the wave ends
}

int main() {
    int *out = nullptr;
    const int blocks = 512;
    hipMalloc(&out, sizeof(int) * 4 * blocks);
    hipMemset(out, 0xff, sizeof(int) * 4 * blocks);
    hipLaunchKernelGGL(leave_one_by_one, dim3(blocks), dim3(256), 0, 0, out);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed\n");
        return 1;
    }
    int host[4 * blocks];
    hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
    long long bad = 0;
    for (int i = 0; i < 4 * blocks; ++i) {
        bad += host[i];
    }
    printf("%d workgroups x 4 waves leaving after 100 / 200 / 300 / 1000 rounds: %lld rounds with a wrong count (0 = barriers count surviving waves only)\n", blocks, bad);
    return bad != 0;
}
