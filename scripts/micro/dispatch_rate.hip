// Workgroup dispatch cost on MI355X: time of kernels whose workgroups do (almost) nothing, as a function of the number
// of workgroups, their static LDS and their VGPR footprint.  hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dr
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"

template <int LDS_KB, int REGS>
__global__ __launch_bounds__(256) void k_empty(float *out, int never)
{
    __shared__ float lds[LDS_KB > 0 ? LDS_KB * 256 : 1];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x + i);
    if (never) {                      // keeps LDS and the registers alive without executing anything
        lds[threadIdx.x] = r[0];
        __syncthreads();
        float s = lds[(threadIdx.x * 7) & 255];
#pragma unroll
        for (int i = 0; i < REGS; ++i) s += r[i] * lds[(threadIdx.x + i) & 255];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int LDS_KB, int REGS>
static void run(const char *name, float *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int nblk : {256, 1152, 2304, 4608, 9216, 18432}) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_empty<LDS_KB, REGS>), dim3(nblk), dim3(256), 0, 0, out, 0);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        const int n = 50;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_empty<LDS_KB, REGS>), dim3(nblk), dim3(256), 0, 0, out, 0);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("%-22s blocks %6d  %7.2f us per launch  (%.1f ns per workgroup)\n", name, nblk, ms * 1e3 / n, ms * 1e6 / n / nblk);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 64 << 20);
    run<0, 1>("no LDS, few regs", out);
    run<32, 1>("32 KB LDS, few regs", out);
    run<48, 1>("48 KB LDS, few regs", out);
    run<32, 64>("32 KB LDS, 64+ regs", out);
    return 0;
}
