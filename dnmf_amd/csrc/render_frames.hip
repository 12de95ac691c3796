// Synthetic-video render: the inner double loop of the reference's generate_video
// (WUtils/Simulator.py:66-73 with simulate_cell :197-212).  Per frame t and neuron k the reference
// evaluates amp * exp(-r^2 / (2 shape_std)) over the whole volume in float64, casts the patch to fp32 and
// adds it to the frame in fp32, neurons in index order.  One thread per (voxel, frame) does exactly that;
// terms that are 0 after the fp32 cast (below 2^-150) are skipped, which changes nothing.
#include "common.hpp"

namespace dnmf {

__global__ __launch_bounds__(256) void render_frames_kernel(const float *__restrict__ positions,
                                                            const double *__restrict__ traces, int K, int T_total,
                                                            int t0, int T, Volume vol, double shape_std,
                                                            double r2_cut, float *__restrict__ out, long ldo) {
    extern __shared__ double sm[];  // per neuron: cx, cy, cz, amp
    const int t = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        sm[4 * k + 0] = (double)positions[((long)k * 3 + 0) * T_total + t0 + t];
        sm[4 * k + 1] = (double)positions[((long)k * 3 + 1) * T_total + t0 + t];
        sm[4 * k + 2] = (double)positions[((long)k * 3 + 2) * T_total + t0 + t];
        sm[4 * k + 3] = traces[(long)k * T_total + t0 + t];
    }
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= vol.P) return;
    int x, y, z;
    voxel_xyz(p, vol, x, y, z);
    const double inv = 0.5 / shape_std;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        const double dx = (double)x - sm[4 * k], dy = (double)y - sm[4 * k + 1], dz = (double)z - sm[4 * k + 2];
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < r2_cut) acc += (float)(sm[4 * k + 3] * exp(-inv * r2));
    }
    out[(long)t * ldo + p] = acc;
}

}  // namespace dnmf

extern "C" int dnmf_render_frames(const float *positions, const double *traces, int K, int T_total, int t0, int T,
                                  int X, int Y, int Z, double shape_std, double amp_max, float *out, long ldo,
                                  dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(positions && traces && out, DNMF_E_NULL, "dnmf_render_frames: NULL buffer");
    DNMF_REQUIRE(K > 0 && T > 0 && t0 >= 0 && t0 + T <= T_total && X > 0 && Y > 0 && Z > 0 && shape_std > 0 && T <= 65535,
                 DNMF_E_SHAPE, "dnmf_render_frames: K=%d T_total=%d t0=%d T=%d X=%d Y=%d Z=%d", K, T_total, t0, T, X, Y, Z);
    const Volume vol = make_volume(X, Y, Z);
    DNMF_REQUIRE(ldo >= vol.P, DNMF_E_SHAPE, "dnmf_render_frames: ldo=%ld < P=%ld", ldo, vol.P);
    DNMF_REQUIRE((size_t)K * 32 <= 64 * 1024, DNMF_E_UNSUPPORTED, "dnmf_render_frames: K=%d too large for the LDS table", K);
    // amp*exp(-r2/(2 s)) < 2^-151 rounds to 0 in fp32: r2 > 2 s (ln amp + 151 ln 2)
    const double r2_cut = 2.0 * shape_std * (log(amp_max > 1.0 ? amp_max : 1.0) + 151.0 * 0.6931471805599453) + 1.0;
    const dim3 grid((unsigned)((vol.P + 255) / 256), (unsigned)T);
    hipLaunchKernelGGL(render_frames_kernel, grid, dim3(256), (size_t)K * 32, (hipStream_t)stream, positions, traces, K,
                       T_total, t0, T, vol, shape_std, r2_cut, out, ldo);
    return check_launch("dnmf_render_frames");
}
