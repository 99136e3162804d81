// What HBM rate does a plain elementwise kernel reach on this GPU as a function of HOW MANY arrays it streams at once? The fused
// optimiser step (csrc/norm.hip layer_update_*) reads four fp32 arrays (accumulator, master, m, v) and writes three fp32 (master, m,
// v) plus two 16-bit operand copies; alone it runs at 3.3-4.2 TB/s. This probe times the simplest possible kernels with the same
// stream counts - 16-byte accesses, fully coalesced, no LDS, no transposition - at the size of the critics' head conv (13.2 M
// elements) and of the generator's fc (20 M): the rate THEY reach is the practical roof of that access mix.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

template <int R, int W, int H, bool NT>
__global__ __launch_bounds__(256) void mix(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, const f4* __restrict__ d,
                                           f4* __restrict__ x, f4* __restrict__ y, f4* __restrict__ z, us4* __restrict__ h1, us4* __restrict__ h2,
                                           long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        auto ld = [&](const f4* p) { return NT ? __builtin_nontemporal_load(p + i) : p[i]; };
        f4 v = ld(a);
        if (R > 1) v += ld(b);
        if (R > 2) v += ld(c);
        if (R > 3) v += ld(d);
        auto st = [&](f4* p, f4 q) { if (NT) __builtin_nontemporal_store(q, p + i); else p[i] = q; };
        st(x, v);
        if (W > 1) st(y, v * 2.f);
        if (W > 2) st(z, v * 3.f);
        us4 q = {(unsigned short)(__float_as_uint(v.x) >> 16), (unsigned short)(__float_as_uint(v.y) >> 16),
                 (unsigned short)(__float_as_uint(v.z) >> 16), (unsigned short)(__float_as_uint(v.w) >> 16)};
        if (H > 0) h1[i] = q;
        if (H > 1) h2[i] = q;
    }
}

template <int R, int W, int H, bool NT>
void run(const char* name, float** bufs, unsigned short** hb, long n, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] {
        hipLaunchKernelGGL((mix<R, W, H, NT>), dim3(blocks), dim3(256), 0, 0, (const f4*)bufs[0], (const f4*)bufs[1], (const f4*)bufs[2],
                           (const f4*)bufs[3], (f4*)bufs[4], (f4*)bufs[5], (f4*)bufs[6], (us4*)hb[0], (us4*)hb[1], n / 4);
    };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n * (4.0 * (R + W) + 2.0 * H);
    printf("  %-34s %8.1f us   %6.2f TB/s\n", name, ms * 1000.0 / reps, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    for (long n : {13222080L, 20086784L, 134217728L}) {
        float* bufs[7];
        unsigned short* hb[2];
        for (auto& b : bufs) { hipMalloc(&b, n * 4); hipMemset(b, 0, n * 4); }
        for (auto& b : hb) { hipMalloc(&b, n * 2); hipMemset(b, 0, n * 2); }
        for (int blocks : {2048, 8192}) {
            printf("n = %ld elements per array, %d blocks of 256\n", n, blocks);
            run<1, 1, 0, false>("1 read, 1 write (copy)", bufs, hb, n, blocks);
            run<1, 1, 0, true>("1 read, 1 write, non-temporal", bufs, hb, n, blocks);
            run<2, 1, 0, false>("2 reads, 1 write", bufs, hb, n, blocks);
            run<4, 1, 0, false>("4 reads, 1 write", bufs, hb, n, blocks);
            run<4, 3, 0, false>("4 reads, 3 writes", bufs, hb, n, blocks);
            run<4, 3, 2, false>("4 reads, 3 writes + 2 16-bit writes", bufs, hb, n, blocks);
            run<4, 3, 2, true>("the same, non-temporal", bufs, hb, n, blocks);
            run<1, 3, 0, false>("1 read, 3 writes", bufs, hb, n, blocks);
        }
        for (auto& b : bufs) hipFree(b);
        for (auto& b : hb) hipFree(b);
    }
    return 0;
}
