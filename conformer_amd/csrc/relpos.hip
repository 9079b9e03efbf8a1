// Relative positional encoding table (2T-1, d), computed once per T (no batch repeat).
#include "cfm_common.h"

__global__ __launch_bounds__(256) void relpos_table_kernel(const float* __restrict__ div_term, float* __restrict__ pe,
                                                           int T, int d) {
    const int half = d >> 1;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)(2 * T - 1) * half;
    if (idx >= total) return;
    const int j = (int)(idx / half), c = (int)(idx % half);
    const int r = T - 1 - j;
    // position.py:14-20 forms angle = |r| * div_term in fp32, then sin/cos(+angle) or sin/cos(-1*angle)
    float ang = (float)(r < 0 ? -r : r) * div_term[c];
    if (r < 0) ang = -ang;
    float2 sc;
    sc.x = sinf(ang);
    sc.y = cosf(ang);
    *reinterpret_cast<float2*>(pe + (int64_t)j * d + 2 * c) = sc;
}

extern "C" int cfm_relpos_table_f32(const float* div_term, float* pe, int T, int d, cfm_stream_t stream) {
    CFM_REQUIRE(div_term && pe, CFM_ERR_NULL);
    CFM_REQUIRE(T > 0 && d > 0 && (d & 1) == 0, CFM_ERR_BAD_SHAPE);
    const int64_t total = (int64_t)(2 * T - 1) * (d / 2);
    hipLaunchKernelGGL(relpos_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), div_term, pe, T, d);
    return cfm_launch_status();
}
