// Diagnostic only: what does ds_read_b64_tr_b16 return? LDS[i] = i (16-bit); lane l reads at byte address addr[l].
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void tr_probe(const int* addr, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)((char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
extern "C" int tr_probe_launch(const int* addr, short* out, void* stream) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
    return (int)hipGetLastError();
}
