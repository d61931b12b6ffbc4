// What store patterns reach on MI355X, for the egocentric observation's output: N images of P bytes, zero-filled.
//   A  linear: every thread 16 bytes, grid-stride over the whole buffer (what a device memset does)
//   B  a wave per image, images interleaved over the resident waves, 16 B per lane (ego_sparse_kernel's fill)
//   C  as B, with s_waitcnt vmcnt(0) after every image (ego_sparse_kernel waits before it patches)
//   D  a 256-thread workgroup per image
//   E  as B with 4-byte stores
// hipcc --offload-arch=gfx950 -O3 tools/fill_rate.hip -o tools/fill_rate && tools/fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void fill_linear(uint8_t* out, int64_t bytes)
{
    u32x4* q = (u32x4*)out;
    const int64_t n = bytes >> 4;
    const u32x4 z = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) q[i] = z;
}

template <int MODE>   // 0 = B, 1 = C, 2 = E
__global__ void __launch_bounds__(512) fill_wave_per_image(uint8_t* out, int64_t n_images, int64_t P)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * waves;
    const u32x4 z = {0, 0, 0, 0};
    for (int64_t img = (int64_t)blockIdx.x * waves + wave; img < n_images; img += stride) {
        uint8_t* image = out + img * P;
        const int head = (int)((16u - (uint32_t)(uintptr_t)image) & 15u);
        const int64_t body = (P - head) >> 4;
        const int tail = (int)(P - head - (body << 4));
        if (lane < head) image[lane] = 0;
        if (MODE == 2) {
            uint32_t* w = (uint32_t*)(image + head);
            for (int64_t c = lane; c < 4 * body; c += 64) w[c] = 0;
        } else {
            u32x4* q = (u32x4*)(image + head);
            for (int64_t c = lane; c < body; c += 64) q[c] = z;
        }
        if (lane < tail) image[head + (body << 4) + lane] = 0;
        if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0F70);
    }
}

__global__ void __launch_bounds__(256) fill_block_per_image(uint8_t* out, int64_t n_images, int64_t P)
{
    const u32x4 z = {0, 0, 0, 0};
    for (int64_t img = blockIdx.x; img < n_images; img += gridDim.x) {
        uint8_t* image = out + img * P;
        const int head = (int)((16u - (uint32_t)(uintptr_t)image) & 15u);
        const int64_t body = (P - head) >> 4;
        const int tail = (int)(P - head - (body << 4));
        if ((int)threadIdx.x < head) image[threadIdx.x] = 0;
        u32x4* q = (u32x4*)(image + head);
        for (int64_t c = threadIdx.x; c < body; c += 256) q[c] = z;
        if ((int)threadIdx.x < tail) image[head + (body << 4) + threadIdx.x] = 0;
    }
}

int main()
{
    const int64_t n = 65536, P = 133 * 117, bytes = n * P;
    uint8_t* out;
    hipMalloc((void**)&out, bytes + 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int k = 0; k < 3; ++k) launch();
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    time("hipMemsetAsync", [&] { hipMemsetAsync(out, 0, bytes, 0); });
    time("A linear, 1024 x 256 threads", [&] { fill_linear<<<1024, 256>>>(out, bytes); });
    time("A linear, 65536 x 256 threads", [&] { fill_linear<<<65536, 256>>>(out, bytes); });
    for (int wg : {256, 512, 1024}) {
        char nm[96];
        snprintf(nm, sizeof nm, "B wave per image, %d x 8 waves", wg);
        time(nm, [&] { fill_wave_per_image<0><<<wg, 512>>>(out, n, P); });
        snprintf(nm, sizeof nm, "C   + vmcnt(0) per image, %d x 8 waves", wg);
        time(nm, [&] { fill_wave_per_image<1><<<wg, 512>>>(out, n, P); });
    }
    time("E wave per image, dword stores, 512 x 8", [&] { fill_wave_per_image<2><<<512, 512>>>(out, n, P); });
    time("D workgroup per image, 2048 x 256", [&] { fill_block_per_image<<<2048, 256>>>(out, n, P); });
    time("D workgroup per image, 8192 x 256", [&] { fill_block_per_image<<<8192, 256>>>(out, n, P); });
    time("B wave per image, P = 15616 (aligned)", [&] { fill_wave_per_image<0><<<512, 512>>>(out, n - 300, 15616); });
    return 0;
}
