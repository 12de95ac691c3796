// Reconstruction image S[b,p] = sum_k C[k,t_b] A[p,k] on the fp32 matrix cores.
//
// The reference reconstructs a frame as einsum('tkmnz,kt->tmnz', A_t, C) AFTER warping all K
// footprints (Demix/dNMF.py:56-58).  The trilinear gather is linear in the footprints, so the same
// image is the gather of S = A.C; K2 then warps ONE channel instead of K.  S is a (P x K).(K x B)
// product: v_mfma_f32_16x16x4_f32, voxels on the rows, frames on the columns.
#include "common.hpp"

namespace dnmf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave owns 16*NF frames and walks `groups_per_wave` groups of 16 voxels.
//   MFMA A operand (16 voxels x 4 k):  lane l -> voxel l&15, k-slot l>>4
//   MFMA B operand (4 k x 16 frames):  lane l -> k-slot l>>4, frame l&15
// The reduction index is permuted so that k-slot q covers the contiguous channels [q*Kq, (q+1)*Kq)
// (Kq = Kp/4): every lane then reads its footprint row with 16-byte loads.
template <int NB, int NF>
__global__ __launch_bounds__(256) void recon_image_kernel(const float *__restrict__ Apk, long P, int K, int Kp,
                                                          const float *__restrict__ C, long ldc,
                                                          const int *__restrict__ times, int B,
                                                          float *__restrict__ S, long lds, int groups_per_wave,
                                                          Volume vol, HaloLayout hl) {
    constexpr int KQ = 4 * NB;  // channels per k-slot
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vi = lane & 15, q = lane >> 4;
    const long ngroups = (P + 15) / 16;
    const long g_begin = ((long)blockIdx.x * 4 + wave) * groups_per_wave;
    if (g_begin >= ngroups) return;
    const long g_end = g_begin + groups_per_wave < ngroups ? g_begin + groups_per_wave : ngroups;
    const int f0 = blockIdx.y * 16 * NF;

    // traces of this wave's frames: bq[nf][s] = C[q*KQ + s][t(f0 + 16 nf + vi)]
    float bq[NF][KQ];
    int tcol[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int f = f0 + 16 * nf + vi;
        tcol[nf] = times[f < B ? f : B - 1];
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            const int ch = q * KQ + s;
            bq[nf][s] = ch < K ? C[(long)ch * ldc + tcol[nf]] : 0.0f;
        }
    }

    // footprint rows of the first group; inside the loop the next group's rows are requested before the MFMAs of the
    // current one so that their latency hides behind 4*NB*NF matrix instructions
    auto load_rows = [&](long g, f32x4 (&dst)[NB]) {
        const long p0 = g * 16;
        const long prow = p0 + vi < P ? p0 + vi : P - 1;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Apk + prow * Kp + q * KQ);
#pragma unroll
        for (int i = 0; i < NB; ++i) dst[i] = src[i];
    };
    f32x4 a4[NB], a4n[NB];
    load_rows(g_begin, a4);
    for (long g = g_begin; g < g_end; ++g) {
        const long p0 = g * 16;
        load_rows(g + 1 < g_end ? g + 1 : g, a4n);
        f32x4 acc[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], bq[nf][4 * i + e], acc[nf], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) a4[i] = a4n[i];
        // D[row = voxel 4q + r][col = frame vi]
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int f = f0 + 16 * nf + vi;
            if (f < B) {
                // halo layout: voxel (x, y, z) lands at (x + HALO) rowf + (y + HALO) Z + z
                float *dst = S + (long)f * lds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long pr = p0 + 4 * q + r;
                    if (pr < P) {
                        int x, y, z;
                        voxel_xyz(pr, vol, x, y, z);
                        dst[(long)(x + HALO) * hl.rowf + (long)(y + HALO) * vol.Z + z] = acc[nf][r];
                    }
                }
            }
        }
    }
}

// zero border of B halo-layout images: the first and last HALO rows whole; of the other rows the HALO columns in front
// and everything behind the volume (HALO columns and the alignment excess)
__global__ __launch_bounds__(256) void halo_zero_kernel(float *__restrict__ S, long lds, int X, int Y, int Z, int rowf) {
    const long front = (long)HALO * Z, back = rowf - (long)(HALO + Y) * Z, edge = front + back;
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    float *dst = S + (long)blockIdx.y * lds;
    if (j < 2L * HALO * rowf) {
        const long row = j / rowf, off = j - row * rowf;
        dst[(row < HALO ? row : X + row) * rowf + off] = 0.0f;
    } else if (j < 2L * HALO * rowf + (long)X * edge) {
        const long jj = j - 2L * HALO * rowf, x = jj / edge, e = jj - x * edge;
        dst[(x + HALO) * rowf + (e < front ? e : rowf - edge + e)] = 0.0f;
    }
}

template <int NB>
static int launch_recon(const float *Apk, const Volume &vol, const HaloLayout &hl, int K, int Kp, const float *C, long ldc,
                        const int *times, int B, float *S, long lds, hipStream_t stream) {
    const long P = vol.P;
    const long ngroups = (P + 15) / 16;
    if (B > 16) {
        constexpr int NF = 4;
        const int fblocks = (B + 16 * NF - 1) / (16 * NF);
        const int gpw = 64;
        const long nwg = (ngroups + 4L * gpw - 1) / (4L * gpw);
        hipLaunchKernelGGL((recon_image_kernel<NB, NF>), dim3((unsigned)nwg, (unsigned)fblocks), dim3(256), 0, stream,
                           Apk, P, K, Kp, C, ldc, times, B, S, lds, gpw, vol, hl);
    } else {
        const int gpw = 16;
        const long nwg = (ngroups + 4L * gpw - 1) / (4L * gpw);
        hipLaunchKernelGGL((recon_image_kernel<NB, 1>), dim3((unsigned)nwg, 1u), dim3(256), 0, stream, Apk, P, K, Kp,
                           C, ldc, times, B, S, lds, gpw, vol, hl);
    }
    return check_launch("dnmf_recon_image");
}

}  // namespace dnmf

extern "C" int dnmf_recon_image(const float *Apk, int X, int Y, int Z, int K, int Kp, const float *C, long ldc,
                                const int *times, int B, float *S, long lds, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(Apk && C && times && S, DNMF_E_NULL, "dnmf_recon_image: NULL buffer");
    DNMF_REQUIRE(X > 0 && Y > 0 && Z > 0 && K > 0 && B > 0 && B <= 65535 && Kp == dnmf_padded_k(K) && ldc > 0, DNMF_E_SHAPE,
                 "dnmf_recon_image: X=%d Y=%d Z=%d K=%d Kp=%d B=%d ldc=%ld", X, Y, Z, K, Kp, B, ldc);
    const Volume vol = make_volume(X, Y, Z);
    const HaloLayout hl = make_halo_layout(X, Y, Z);
    DNMF_REQUIRE(lds >= hl.Pp, DNMF_E_SHAPE, "dnmf_recon_image: lds=%ld < %ld floats of a halo-layout image", lds, hl.Pp);
    DNMF_REQUIRE((reinterpret_cast<size_t>(Apk) & 15) == 0, DNMF_E_SHAPE, "dnmf_recon_image: Apk must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long nborder = hl.Pp - vol.P;
    hipLaunchKernelGGL(halo_zero_kernel, dim3((unsigned)((nborder + 255) / 256), (unsigned)B), dim3(256), 0, st, S, lds, X, Y,
                       Z, hl.rowf);
    switch (Kp / 16) {
        case 1: return launch_recon<1>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 2: return launch_recon<2>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 3: return launch_recon<3>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 4: return launch_recon<4>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 5: return launch_recon<5>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 6: return launch_recon<6>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 7: return launch_recon<7>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        case 8: return launch_recon<8>(Apk, vol, hl, K, Kp, C, ldc, times, B, S, lds, st);
        default:
            return fail(DNMF_E_UNSUPPORTED, "dnmf_recon_image: K=%d needs Kp=%d > 128 (not built yet)", K, Kp);
    }
}
