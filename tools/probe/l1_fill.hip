// Does an LDS-DMA fill (global_load_lds dwordx4) run faster when its source lines are resident in the CU's vector L1 than when they
// come from L2?  One block of 256 threads per CU fills a 32 KB LDS buffer over and over from (a) the same 16 KB (L1-resident after the
// first trip), (b) a 512 KB window private to the block (L2-resident, 16x the L1), (c) like (b) but 8 rows x 128 B per wave instruction
// at a 2 KB row stride (the gather-GEMM's A pattern). Prints bytes per clock and CU.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/l1_fill.hip -o /tmp/l1_fill && /tmp/l1_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void fill(const unsigned char* src, long window, long row_stride, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (long)blockIdx.x * window;
    const long lane_off = row_stride ? (long)(lane >> 3) * row_stride + (lane & 7) * 16 : (long)lane * 16;
    const long inst_bytes = row_stride ? 8 * row_stride : 1024;
    long pos = (long)wave * inst_bytes;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {            // 8 instructions x 1 KB per wave and trip = 32 KB per block
            const unsigned char* p = base + pos + lane_off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(smem + (wave * 8 + k) * 1024), 16, 0, 0);
            pos += 4 * inst_bytes;
            if (pos + inst_bytes > window) pos = (long)wave * inst_bytes;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 2000;
    const long big = 512 << 10;
    unsigned char* src; unsigned long long* cyc;
    hipMalloc(&src, (size_t)blocks * big + (1 << 20)); hipMemset(src, 1, (size_t)blocks * big + (1 << 20));
    hipMalloc(&cyc, blocks * 8);
    struct { const char* name; long window, stride; } cases[] = {
        {"16 KB window, linear (L1-resident)", 16 << 10, 0}, {"32 KB window, linear", 32 << 10, 0}, {"512 KB window, linear (L2)", big, 0},
        {"16 KB window, rows of 128 B at 2 KB stride... (8 rows span 16 KB)", 16 << 10, 2048}, {"512 KB window, rows of 128 B at 2 KB stride (L2)", big, 2048}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 32 << 10, 0, src, c.window, c.stride, iters, cyc);
            hipDeviceSynchronize();
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 32 << 10, 0, src, c.window, c.stride, iters, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double bytes = 32768.0 * iters;
        printf("%-70s %6.1f B/clk/CU  %6.2f TB/s chip  (%.2f GHz)\n", c.name, bytes / avg, bytes * blocks / (ms * 1e-3) / 1e12, avg / (ms * 1e6));
    }
    return 0;
}
