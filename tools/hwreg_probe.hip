// Where does the dispatcher put the waves of a workgroup, and which workgroups share a compute unit?  step_local_kernel's
// waves have unequal work (mover > scorer > helper) and a SIMD's issue port is the step's binding resource, so the roles of
// co-resident workgroups must land on different SIMDs.  Every wave records HW_ID, XCC_ID, LDS_ALLOC,
// GPR_ALLOC and s_memtime; the host prints, for workgroups of 16 / 8 / 4 waves with step_local_kernel's LDS sizes,
//   * the SIMD of wave w of a workgroup (is it w % 4 ?),
//   * for each CU the workgroups resident together at the start, their blockIdx and LDS bases,
//   * how a second round of workgroups is placed.
//   hipcc --offload-arch=gfx950 -O3 tools/hwreg_probe.hip -o tools/hwreg_probe && tools/hwreg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#include <algorithm>

extern __shared__ uint32_t lds[];

__global__ void probe(uint32_t* out, int spin)
{
    uint32_t hw_id, xcc, lds_alloc, gpr_alloc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_GPR_ALLOC)" : "=s"(gpr_alloc));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t acc = lds[(threadIdx.x * 7) % blockDim.x];
    for (int k = 0; k < spin; ++k) acc = acc * 1664525u + 1013904223u;   // keep the workgroup resident for a while
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        uint32_t* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;
        o[0] = hw_id; o[1] = xcc; o[2] = lds_alloc; o[3] = gpr_alloc;
        o[4] = (uint32_t)t0; o[5] = (uint32_t)(t0 >> 32); o[6] = (uint32_t)(t1 - t0); o[7] = acc;
    }
}

int main()
{
    const int configs[3][3] = {{16, 47 * 1024, 256}, {8, 26 * 1024, 512}, {4, 15 * 1024, 1024}};   // waves, LDS bytes, workgroups
    for (int rounds = 1; rounds <= 2; ++rounds)
    for (auto& cfg : configs) {
        const int waves = cfg[0], lds_bytes = cfg[1], groups = cfg[2] * rounds;
        uint32_t* out;
        hipMalloc(&out, (size_t)groups * waves * 8 * 4);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(probe, dim3(groups), dim3(64 * waves), lds_bytes, 0, out, 20000);
        hipDeviceSynchronize();
        std::vector<uint32_t> h((size_t)groups * waves * 8);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        hipFree(out);
        printf("==== %d waves per workgroup, %d B of LDS, %d workgroups\n", waves, lds_bytes, groups);
        // SIMD of wave w
        int simd_hist[16][4] = {};
        for (int b = 0; b < groups; ++b)
            for (int w = 0; w < waves; ++w) simd_hist[w][(h[((size_t)b * waves + w) * 8] >> 4) & 3]++;
        for (int w = 0; w < waves; ++w)
            printf("  wave %2d on SIMD 0/1/2/3: %d %d %d %d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
        // who shares a CU: key = (xcc, se, sh, cu)
        std::map<uint32_t, std::vector<int>> per_cu;
        for (int b = 0; b < groups; ++b) {
            const uint32_t id = h[(size_t)b * waves * 8], xcc = h[(size_t)b * waves * 8 + 1] & 0xF;
            const uint32_t key = (xcc << 16) | (id & 0xFF00);   // cu_id [11:8], sh_id [12], se_id [15:13]
            per_cu[key].push_back(b);
        }
        printf("  %zu distinct CUs in use\n", per_cu.size());
        int shown = 0;
        for (auto& kv : per_cu) {
            if (shown++ >= 6) break;
            printf("  CU %05x:", kv.first);
            for (int b : kv.second) {
                const uint32_t* o = &h[(size_t)b * waves * 8];
                printf("  [block %4d lds_alloc %08x gpr_alloc %08x wave0: hw_id %08x start %llu]", b, o[2], o[3], o[0],
                       (unsigned long long)(((uint64_t)o[5] << 32) | o[4]) % 100000000ull);
            }
            printf("\n");
        }
        // does block b's CU partner differ by a fixed stride?
        std::map<int, int> stride_hist;
        for (auto& kv : per_cu)
            for (size_t k = 1; k < kv.second.size(); ++k) stride_hist[kv.second[k] - kv.second[k - 1]]++;
        printf("  differences between the block indices sharing a CU:");
        for (auto& kv : stride_hist) printf(" %d x%d", kv.first, kv.second);
        printf("\n");
    }
    return 0;
}
