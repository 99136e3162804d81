// Semantics of buffer_load_dwordx4 ... lds on gfx950 as the gather-GEMM would use it: (1) where do the 64 lanes' 16 bytes land in LDS,
// (2) does a lane whose voffset is >= num_records deliver ZEROS (the zero padding of a conv without a zero page and without a
// select on a 64-bit pointer), (3) is the scalar offset part of the address but NOT of the range check (so the per-K-tile advance
// can live in an SGPR), (4) may the resource base lie below the tensor (negative tap offsets folded into a bias).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/buf_lds.hip -o /tmp/buf_lds && /tmp/buf_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* A, float* out, unsigned soff, int below) {
    __shared__ __attribute__((aligned(16))) float sm[4 * 256];
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) sm[i] = -1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(A) - (below ? 4096 : 0);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000u, 0x00020000);
    unsigned voff = (unsigned)(lane * 16 + wave * 8192);
    if (lane % 3 == 1) voff = 0x80000000u + lane * 16;          // out of range: must arrive as zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, voff,
                                             soff + (below ? 4096 : 0), 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) out[i] = sm[i];
}

int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *A, *out;
    hipMalloc(&A, N * 4 + 8192); hipMalloc(&out, 4096 * 4);
    A += 1024;                                                   // (room below the tensor for the `below` case)
    hipMemcpy(A, h.data(), N * 4, hipMemcpyHostToDevice);
    for (int below = 0; below < 2; ++below)
        for (unsigned soff : {0u, 1024u, 1u << 20}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, A, out, soff, below);
            std::vector<float> o(1024);
            hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
            int bad = 0, zeros = 0;
            for (int w = 0; w < 4; ++w)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const float got = o[w * 256 + l * 4 + e];
                        const float want = (l % 3 == 1) ? 0.f : (float)((soff + l * 16 + w * 8192) / 4 + e);
                        if (got != want) { if (bad < 4) printf("   wave %d lane %d e %d: got %g want %g\n", w, l, e, got, want); ++bad; }
                        if (l % 3 == 1 && got == 0.f) ++zeros;
                    }
            printf("base %s tensor, soffset %u: %d mismatches, %d zero-filled elements of %d out-of-range ones\n", below ? "4096 B below the" : "at the",
                   soff, bad, zeros, 4 * 21 * 4);
        }
    return 0;
}
