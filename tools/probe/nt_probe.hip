// Stand-alone timing of gemm_nt_kernel variants (CPCSV_PROBE bitmask) on dense shapes: where does a block's fixed
// cost go?   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DCPCSV_PROBE=<mask> -I include
//            tools/probe/nt_probe.hip -o nt_probe_<mask>
#include "../../cpcstoryvisualization-pytorch_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

static void run(int M, int N, int K, int reps) {
    void *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 2);
    hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C, (size_t)M * N * 2);
    std::vector<uint16_t> h((size_t)M * K > (size_t)N * K ? (size_t)M * K : (size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)(rand() & 0xff);
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    cpcsv_gemm_desc d = {};
    d.A = A; d.B = B; d.C = C; d.dtype = CPCSV_BF16; d.M = M; d.N = N; d.Cs = K; d.ldb = K; d.ldc = N;
    d.ntaps = 1; d.MH = d.MW = d.IH = d.IW = 1; d.sy = d.sx = 1; d.splitk = 1; d.nphases = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) cpcsv_gemm_nt(&d, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) cpcsv_gemm_nt(&d, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("probe=%d M=%6d N=%5d K=%6d  %8.1f us\n", CPCSV_PROBE, M, N, K, ms / reps * 1e3);
#if CPCSV_PROBE & 8
    unsigned long long hp[8], z[8] = {0};
    hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_probe), sizeof(hp));
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z));
    const double nb = (double)hp[4], nk = (K + 63) / 64;
    printf("   wave0 per block: prologue+loop %.0f cyc; per K tile: issue %.0f  mma %.0f  wait+barrier %.0f  (blocks %.0f)\n",
           hp[3] / nb, hp[0] / nb / nk, hp[1] / nb / nk, hp[2] / nb / nk, nb / (reps + 3));
#endif
    hipFree(A); hipFree(B); hipFree(C);
}

int main(int argc, char** argv) {
    if (argc > 4) {
        run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
        return 0;
    }
    run(8192, 2048, 64, 50);
    run(8192, 2048, 512, 50);
    run(8192, 2048, 4096, 20);
    run(960, 992, 7936, 50);
    return 0;
}
