// vec_kernels.hip — HBM-bound O(N*n) kernels of the ChASE hot path (wave64 shuffle reductions, 16-byte accesses)
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "ctx.h"
#include "../../include/chase_hip.h"

int chase_hip_ctx::ensure_ws(size_t bytes)
{
    if (ws_bytes >= bytes) return 0;
    if (ws) { hipStreamSynchronize(stream); hipFree(ws); ws = nullptr; ws_bytes = 0; }
    hipError_t e = hipMalloc(&ws, bytes);
    if (e != hipSuccess) return chase_hip::set_error(CHASE_HIP_ENOMEM, "workspace allocation failed");
    ws_bytes = bytes;
    return 0;
}

namespace chase_hip {

__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

int stream_copy(hipStream_t st, void* dst, const void* src, size_t bytes)
{
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, st, (float4*)dst, (const float4*)src, n16);
    return (int)hipGetLastError();
}

} // namespace chase_hip
