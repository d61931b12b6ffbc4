// How fast does the chip start workgroups?  The step kernel's time at 65 536 envs fits  T = 8.8 us + 0.09 us x (workgroups per
// XCD)  for workgroups of 16, 8 and 4 waves alike (profiles/r04_n_sweep_pairs.txt) -- as if starting a workgroup cost ~90 ns
// per XCD whatever its size.  This tool measures it directly: a kernel whose workgroups stamp s_memtime on entry and leave
// after `spin` iterations, for grids of G workgroups of W waves with L bytes of LDS and ~V vector registers; per XCD the
// entry stamps are sorted and the cadence of the first round of workgroups is printed, and the launch time from HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o tools/dispatch_rate && tools/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

extern __shared__ uint32_t lds[];

template <int VREGS>
__global__ void probe(uint64_t* out, int spin, uint32_t* sink)
{
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t v[VREGS];
#pragma unroll
    for (int k = 0; k < VREGS; ++k) v[k] = threadIdx.x * (k + 1);
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int k = 0; k < VREGS; ++k) v[k] = v[k] * 1664525u + v[(k + 1) % VREGS];
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < VREGS; ++k) acc ^= v[k];
    if (acc == 0x12345678u) sink[threadIdx.x] = acc + lds[threadIdx.x];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t0;
        out[2 * blockIdx.x + 1] = ((uint64_t)(xcc & 0xF) << 56) | (__builtin_amdgcn_s_memtime() - t0);
    }
}

template <int VREGS>
static void run(int groups, int waves, int lds_bytes, int spin)
{
    uint64_t* out;
    uint32_t* sink;
    hipMalloc(&out, (size_t)groups * 16);
    hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void*)probe<VREGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe<VREGS>, dim3(groups), dim3(64 * waves), lds_bytes, 0, out, spin, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 100;
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(probe<VREGS>, dim3(groups), dim3(64 * waves), lds_bytes, 0, out, spin, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h((size_t)groups * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    // cadence per XCD: sorted entry stamps
    double cadence = 0, span = 0, body = 0;
    int xcds = 0;
    for (int x = 0; x < 8; ++x) {
        std::vector<uint64_t> t;
        for (int b = 0; b < groups; ++b)
            if ((int)(h[2 * b + 1] >> 56) == x) {
                t.push_back(h[2 * b]);
                body += (double)(h[2 * b + 1] & 0xFFFFFFFFFFFFull);
            }
        if (t.size() < 2) continue;
        std::sort(t.begin(), t.end());
        span += (double)(t.back() - t.front());
        cadence += (double)(t.back() - t.front()) / (double)(t.size() - 1);
        ++xcds;
    }
    printf("%5d workgroups x %2d waves, %6d B LDS, ~%3d VGPRs, spin %5d: %8.3f us per launch; per XCD: first-to-last entry %7.0f cycles, "
           "%6.1f cycles between entries; body %6.0f cycles\n", groups, waves, lds_bytes, VREGS + 8, spin, ms * 1e3 / reps,
           span / std::max(xcds, 1), cadence / std::max(xcds, 1), body / groups);
    hipFree(out);
    hipFree(sink);
}

int main()
{
    const int spins[2] = {0, 300};
    for (int spin : spins) {
        printf("---- spin %d\n", spin);
        for (int lds_bytes : {0, 16 * 1024, 47 * 1024}) {
            run<16>(256, 16, lds_bytes, spin);
            run<16>(512, 8, lds_bytes > 32 * 1024 ? 26 * 1024 : lds_bytes, spin);
            run<16>(1024, 4, lds_bytes > 32 * 1024 ? 15 * 1024 : lds_bytes, spin);
        }
        run<96>(256, 16, 47 * 1024, spin);
        run<96>(512, 8, 26 * 1024, spin);
        run<96>(1024, 4, 15 * 1024, spin);
        run<96>(128, 16, 47 * 1024, spin);
        run<96>(64, 16, 47 * 1024, spin);
        run<96>(2048, 4, 15 * 1024, spin);
        run<16>(256, 4, 0, spin);
        run<16>(2048, 4, 0, spin);
        run<16>(4096, 4, 0, spin);
        run<16>(4096, 1, 0, spin);
    }
    return 0;
}
