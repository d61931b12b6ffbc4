// LDS read throughput per CU by instruction width and address pattern (gfx950): cycles per wave64 instruction with
// 16 waves of one workgroup reading 2048 times each.   hipcc --offload-arch=gfx950 -O3 tools/lds_rate.hip -o tools/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__shared__ uint32_t lds[12 * 1024];

#define REP8(x) x x x x x x x x
#define KERNEL(NAME, INSTR, REG)                                                                                     \
    __global__ void __launch_bounds__(1024) NAME(const uint32_t* addr_in, uint32_t* out, uint64_t* cycles)          \
    {                                                                                                                \
        for (int k = threadIdx.x; k < 12 * 1024; k += blockDim.x) lds[k] = k;                                        \
        uint32_t a = addr_in[threadIdx.x & 63];                                                                      \
        REG r0, r1, r2, r3, r4, r5, r6, r7;                                                                          \
        __syncthreads();                                                                                             \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                            \
        for (int it = 0; it < 256; ++it) {                                                                           \
            asm volatile(INSTR " %0, %8\n" INSTR " %1, %8 offset:8\n" INSTR " %2, %8 offset:16\n" INSTR               \
                         " %3, %8 offset:24\n" INSTR " %4, %8 offset:32\n" INSTR " %5, %8 offset:40\n" INSTR           \
                         " %6, %8 offset:48\n" INSTR " %7, %8 offset:56\ns_waitcnt lgkmcnt(0)\n"                      \
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)             \
                         : "v"(a));                                                                                  \
        }                                                                                                            \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                            \
        if ((threadIdx.x & 63) == 0) {                                                                               \
            cycles[2 * (threadIdx.x >> 6)] = t0;                                                                     \
            cycles[2 * (threadIdx.x >> 6) + 1] = t1;                                                                 \
        }                                                                                                            \
        out[threadIdx.x] = (uint32_t)(uint64_t)r0 ^ (uint32_t)(uint64_t)r7;                                          \
    }
typedef uint32_t u32;
typedef uint64_t u64;
KERNEL(k_u8, "ds_read_u8", u32)
KERNEL(k_u8hi, "ds_read_u8_d16_hi", u32)
KERNEL(k_u16, "ds_read_u16", u32)
KERNEL(k_b32, "ds_read_b32", u32)
KERNEL(k_b64, "ds_read_b64", u64)

int main()
{
    uint32_t *addr, *out;
    uint64_t* cyc;
    hipMalloc(&addr, 64 * 4);
    hipMalloc(&out, 1024 * 4);
    hipMalloc(&cyc, 64 * 8);
    struct { const char* name; void (*fn)(const uint32_t*, uint32_t*, uint64_t*); int align; } ks[] = {
        {"ds_read_u8", k_u8, 1}, {"ds_read_u8_d16_hi", k_u8hi, 1}, {"ds_read_u16", k_u16, 2}, {"ds_read_b32", k_b32, 4},
        {"ds_read_b64", k_b64, 8}, {"ds_read_b64 (+4 B)", k_b64, -8}};
    const char* patterns[] = {"same address", "consecutive bytes", "consecutive dwords", "8 bytes apart", "random",
                              "rotated rows (pitch 185)", "half the lanes (random)"};
    printf("%-20s", "cycles / instr / CU");
    for (auto p : patterns) printf(" %24s", p);
    printf("\n");
    for (auto& k : ks) {
        printf("%-20s", k.name);
        for (int p = 0; p < 7; ++p) {
            uint32_t h[64];
            uint32_t seed = 12345;
            for (int l = 0; l < 64; ++l) {
                seed = seed * 1664525u + 1013904223u;
                uint32_t a = 0;
                switch (p) {
                    case 0: a = 1024; break;
                    case 1: a = 1024 + l; break;
                    case 2: a = 1024 + 4 * l; break;
                    case 3: a = 1024 + 8 * l; break;
                    case 4: case 6: a = (seed >> 8) % 32768; break;
                    case 5: a = 1024 + (l / 16) * 185 * 3 + (uint32_t)((l % 16) * 8 * 0.8) + ((uint32_t)((l % 16) * 8 * 0.6)) * 185; break;
                }
                h[l] = k.align > 0 ? a / k.align * k.align : a / 8 * 8 + 4;   // (negative: 8-byte reads 4 bytes off their alignment)
            }
            hipMemcpy(addr, h, sizeof(h), hipMemcpyHostToDevice);
            uint64_t best = ~0ull;
            for (int rep = 0; rep < 3; ++rep) {
                // (pattern 6: only the kernel's lanes 0..31 matter -- emulated by giving lanes 32..63 lane 0's address)
                if (p == 6) { for (int l = 32; l < 64; ++l) h[l] = h[0]; hipMemcpy(addr, h, sizeof(h), hipMemcpyHostToDevice); }
                hipLaunchKernelGGL(k.fn, dim3(1), dim3(1024), 0, 0, addr, out, cyc);
                uint64_t t[32];
                hipMemcpy(t, cyc, sizeof(t), hipMemcpyDeviceToHost);
                uint64_t first = ~0ull, last = 0;
                for (int w = 0; w < 16; ++w) {
                    first = t[2 * w] < first ? t[2 * w] : first;
                    last = t[2 * w + 1] > last ? t[2 * w + 1] : last;
                }
                if (last - first < best) best = last - first;
            }
            printf(" %24.2f", (double)best / (16.0 * 2048.0));
        }
        printf("\n");
    }
    return 0;
}
