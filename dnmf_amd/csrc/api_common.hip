// Error plumbing, version and the footprint packer of libdnmf_hip.so.
#include "common.hpp"

namespace dnmf {

static thread_local char g_err[512] = "";

char *last_error_buffer() { return g_err; }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return DNMF_OK;
}

// Apk[p][c] = c < K ? A[p][c] : 0
__global__ void pack_footprints_kernel(const float *__restrict__ A, long P, int K, float *__restrict__ Apk, int Kp) {
    const long n = P * Kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long p = i / Kp;
        const int c = (int)(i - p * Kp);
        Apk[i] = c < K ? A[p * K + c] : 0.0f;
    }
}

}  // namespace dnmf

extern "C" {

int dnmf_version(void) { return DNMF_ABI_VERSION; }

const char *dnmf_last_error(void) { return dnmf::last_error_buffer(); }

#ifndef DNMF_BUILD_STAMP
#define DNMF_BUILD_STAMP ""
#endif
const char *dnmf_build_stamp(void) { return DNMF_BUILD_STAMP; }

int dnmf_padded_k(int K) { return K < 1 ? 0 : 16 * ((K + 1 + 15) / 16); }

int dnmf_pack_footprints(const float *A, long P, int K, float *Apk, int Kp, dnmf_stream_t stream) {
    DNMF_REQUIRE(A && Apk, DNMF_E_NULL, "dnmf_pack_footprints: NULL buffer");
    DNMF_REQUIRE(P > 0 && K > 0 && Kp == dnmf_padded_k(K), DNMF_E_SHAPE,
                 "dnmf_pack_footprints: P=%ld K=%d Kp=%d (want Kp=%d)", P, K, Kp, dnmf_padded_k(K));
    const long n = P * Kp;
    const int block = 256;
    const int grid = (int)((n + block - 1) / block < 8192 ? (n + block - 1) / block : 8192);
    hipLaunchKernelGGL(dnmf::pack_footprints_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, A, P, K, Apk, Kp);
    return dnmf::check_launch("dnmf_pack_footprints");
}

}  // extern "C"
