// K1 -- materialised warp: A_t (B,K,X,Y,Z) and the normalised grid (X,Y,Z,3,B).
// Replaces the einsum / normalise / F.grid_sample sequence of ExponentialFP.forward
// (reference Demix/dNMF.py:54-57).  Only the drop-in surface and the tests need A_t in memory; the
// fit itself runs on the fused kernels K2 / K3 that never write it.
#include "common.hpp"

namespace dnmf {

// One thread per (voxel, frame); the K channels of a voxel are produced by walking the footprint rows.
// Corner order and arithmetic follow ATen's grid_sampler_3d (tnw, tne, tsw, tse, bnw, bne, bsw, bse).
__global__ __launch_bounds__(256) void warp_gather_kernel(const float *__restrict__ A, Volume vol, int K,
                                                          const float *__restrict__ beta, int T,
                                                          const int *__restrict__ times, int B,
                                                          float *__restrict__ A_t, float *__restrict__ grid) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= vol.P) return;
    float bt[30];
    load_beta(beta, T, times[b], bt);
    int x, y, z;
    voxel_xyz(p, vol, x, y, z);
    const float xf = (float)x, yf = (float)y, zf = (float)z;
    if (grid) {
        const bool hz = vol.Z > 1;
        grid[(p * 3 + 0) * B + b] = hz ? grid_n<true>(bt, vol, 0, xf, yf, zf) : grid_n<false>(bt, vol, 0, xf, yf, 0.0f);
        grid[(p * 3 + 1) * B + b] = hz ? grid_n<true>(bt, vol, 1, xf, yf, zf) : grid_n<false>(bt, vol, 1, xf, yf, 0.0f);
        grid[(p * 3 + 2) * B + b] = hz ? grid_n<true>(bt, vol, 2, xf, yf, zf) : -1.0f;
    }
    if (!A_t) return;
    const Sample s = make_sample(bt, vol, x, y, z);
    float w[8];
    long row[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
        const int cx = s.x0 + dx, cy = s.y0 + dy, cz = s.z0 + dz;
        const bool ok = in_range(cx, vol.X) && in_range(cy, vol.Y) && in_range(cz, vol.Z);
        const float wc = __fmul_rn(__fmul_rn(dx ? s.wx1 : s.wx0, dy ? s.wy1 : s.wy0), dz ? s.wz1 : s.wz0);
        w[c] = ok ? wc : 0.0f;
        row[c] = ok ? (((long)cx * vol.Y + cy) * vol.Z + cz) * K : 0;
    }
    float *out = A_t + (long)b * K * vol.P + p;
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc = __fadd_rn(acc, __fmul_rn(A[row[c] + k], w[c]));
        out[(long)k * vol.P] = acc;
    }
}

}  // namespace dnmf

extern "C" int dnmf_warp_gather(const float *A, int X, int Y, int Z, int K, const float *beta, int T,
                                const int *times, int B, float *A_t, float *grid, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(A && beta && times, DNMF_E_NULL, "dnmf_warp_gather: NULL input");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && T > 0 && B > 0 && B <= 65535, DNMF_E_SHAPE,
                 "dnmf_warp_gather: X=%d Y=%d Z=%d K=%d T=%d B=%d", X, Y, Z, K, T, B);
    if (!A_t && !grid) return DNMF_OK;
    const Volume vol = make_volume(X, Y, Z);
    const dim3 grid_dim((unsigned)((vol.P + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(warp_gather_kernel, grid_dim, dim3(256), 0, (hipStream_t)stream, A, vol, K, beta, T, times, B,
                       A_t, grid);
    return check_launch("dnmf_warp_gather");
}
