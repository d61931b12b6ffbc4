// pmc_calib -- known-byte-count kernels in the step kernels' own access pattern, to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count
// in your own access pattern").  One element per lane and array, wave-contiguous (struct-of-arrays), N lanes:
//   calib_f64   8 x f64 read, 9 x f64 written       (robot state + min_dist in; the same + reward out)
//   calib_i32   2 x i32 read, 3 x i32 written       (target_idx, current_iter; + err)
//   calib_u8    1 x u8 read,  3 x u8 written        (robot_collided; + done, collided_now)
//   calib_f32x2 1 x float2 read                      (actions)
//   calib_step  all of the above in one kernel       (the algorithmic pattern of one env-step: 81 B in, 87 B out)
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/pmc_calib   and   ... --pmc WRITE_SIZE -- tools/pmc_calib
// (tools/step_pmc.sh does both and turns the counter values into bytes-per-counted-byte factors).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define HIP(call)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                     \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct Arrays {
    double* f64_in[8];
    double* f64_out[9];
    int32_t* i32_in[2];
    int32_t* i32_out[3];
    uint8_t* u8_in[1];
    uint8_t* u8_out[3];
    float2* act;
};

__global__ void __launch_bounds__(128) calib_f64(Arrays a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a.f64_in[k][i];
#pragma unroll
    for (int k = 0; k < 9; ++k) a.f64_out[k][i] = s + k;
}

__global__ void __launch_bounds__(128) calib_i32(Arrays a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = a.i32_in[0][i] + a.i32_in[1][i];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.i32_out[k][i] = s + k;
}

__global__ void __launch_bounds__(128) calib_u8(Arrays a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = a.u8_in[0][i];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.u8_out[k][i] = (uint8_t)(s + k);
}

__global__ void __launch_bounds__(128) calib_f32x2(Arrays a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 c = a.act[i];
    if (c.x == 12345.f && c.y == 54321.f) a.f64_out[0][i] = 1.0;   // (never: keeps the load alive)
}

__global__ void __launch_bounds__(128) calib_step(Arrays a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a.f64_in[k][i];
    const int t = a.i32_in[0][i] + a.i32_in[1][i] + a.u8_in[0][i];
    const float2 c = a.act[i];
    s += c.x + c.y;
#pragma unroll
    for (int k = 0; k < 9; ++k) a.f64_out[k][i] = s + k;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.i32_out[k][i] = t + k;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.u8_out[k][i] = (uint8_t)(t + k);
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : (int64_t)65536 * 64;   // 4 Mi lanes: 340 MB in, 365 MB out
    Arrays a;
    for (auto& p : a.f64_in) HIP(hipMalloc((void**)&p, n * 8));
    for (auto& p : a.f64_out) HIP(hipMalloc((void**)&p, n * 8));
    for (auto& p : a.i32_in) HIP(hipMalloc((void**)&p, n * 4));
    for (auto& p : a.i32_out) HIP(hipMalloc((void**)&p, n * 4));
    for (auto& p : a.u8_in) HIP(hipMalloc((void**)&p, n));
    for (auto& p : a.u8_out) HIP(hipMalloc((void**)&p, n));
    HIP(hipMalloc((void**)&a.act, n * 8));
    for (auto& p : a.f64_in) HIP(hipMemset(p, 0, n * 8));
    for (auto& p : a.i32_in) HIP(hipMemset(p, 0, n * 4));
    for (auto& p : a.u8_in) HIP(hipMemset(p, 0, n));
    HIP(hipMemset(a.act, 0, n * 8));
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(calib_f64, grid, block, 0, 0, a, n);
        hipLaunchKernelGGL(calib_i32, grid, block, 0, 0, a, n);
        hipLaunchKernelGGL(calib_u8, grid, block, 0, 0, a, n);
        hipLaunchKernelGGL(calib_f32x2, grid, block, 0, 0, a, n);
        hipLaunchKernelGGL(calib_step, grid, block, 0, 0, a, n);
        HIP(hipDeviceSynchronize());
    }
    printf("{\"lanes\": %lld, \"bytes\": {\"calib_f64\": [%lld, %lld], \"calib_i32\": [%lld, %lld], \"calib_u8\": [%lld, %lld], "
           "\"calib_f32x2\": [%lld, 0], \"calib_step\": [%lld, %lld]}}\n",
           (long long)n, (long long)(n * 64), (long long)(n * 72), (long long)(n * 8), (long long)(n * 12), (long long)n,
           (long long)(n * 3), (long long)(n * 8), (long long)(n * 81), (long long)(n * 87));
    return 0;
}
