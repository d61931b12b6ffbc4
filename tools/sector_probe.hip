// Fetch granularity of scattered reads (what FETCH_SIZE charges for a 4-byte load that misses L2): every thread reads one
// word of a 256-byte block of its own (blocks visited in a scrambled order), then a second word 32 / 64 / 128 bytes further.
//   hipcc --offload-arch=gfx950 -O3 tools/sector_probe.hip -o tools/sector_probe
//   rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/sector_probe -o p -- tools/sector_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int SECOND>   // byte offset of the second word, 0 = none
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ buf, uint32_t* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = (i * 2654435761u) & (n - 1);   // (n is a power of two: a permutation of the blocks)
    uint32_t v = buf[(size_t)b * 64];
    if (SECOND) v += buf[(size_t)b * 64 + SECOND / 4];
    if (v == 0x12345678u) out[i] = v;   // (never: keeps the loads alive)
}

int main()
{
    const uint32_t n = 1u << 21;   // 2 M blocks of 256 bytes = 512 MB
    uint32_t *buf, *out;
    if (hipMalloc((void**)&buf, (size_t)n * 256) != hipSuccess || hipMalloc((void**)&out, (size_t)n * 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, (size_t)n * 256);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<0>, dim3(n / 256), dim3(256), 0, 0, buf, out, n);
        hipLaunchKernelGGL(probe<32>, dim3(n / 256), dim3(256), 0, 0, buf, out, n);
        hipLaunchKernelGGL(probe<64>, dim3(n / 256), dim3(256), 0, 0, buf, out, n);
        hipLaunchKernelGGL(probe<128>, dim3(n / 256), dim3(256), 0, 0, buf, out, n);
    }
    (void)hipDeviceSynchronize();
    printf("done: %u blocks\n", n);
    return 0;
}
