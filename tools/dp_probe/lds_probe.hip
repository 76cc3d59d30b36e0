// Stand-in for an RCCL collective kernel in the single-GPU contention probe (tools/dp_probe/probe.py): a few
// workgroups that each need `lds_bytes` of LDS and spin for ~`spin` clock ticks.  Not part of the product library.
#include <hip/hip_runtime.h>
extern "C" __global__ void k_lds_probe(float* out, int spin) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    long t0 = clock64();
    float acc = 0.f;
    while (clock64() - t0 < spin) acc += sm[(threadIdx.x * 7) & 255];
    if (acc == -1.f) out[0] = acc;
}
extern "C" int lds_probe_launch(float* out, int blocks, int lds_bytes, int spin, void* stream) {
    static int cur = 0;
    if (lds_bytes > cur) {
        if (hipFuncSetAttribute((const void*)k_lds_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -1;
        cur = lds_bytes;
    }
    k_lds_probe<<<blocks, 256, lds_bytes, (hipStream_t)stream>>>(out, spin);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
