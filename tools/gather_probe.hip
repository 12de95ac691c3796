// What a vector load costs the texture path of a CU, by width and address pattern (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/gather_probe.bin && tools/gather_probe.bin
// Every wave issues ITER x 8 independent loads from a 2 MB table (L2-resident) and adds what it gets; 8 waves per SIMD on
// every CU.  Printed: nanoseconds per wave-instruction per CU and the bytes per clock per CU at 2.4 GHz.  Patterns:
//   row      lane l reads at l * width bytes (one contiguous run per instruction)
//   tile16   16 lanes contiguous (width bytes each), the four 16-lane groups 2,176 bytes apart: K3n's tap pattern
//   shift1   as row, every address 4 bytes off the width's natural alignment (a tap pair that starts at an odd float)
//   lds      the same number of ds_read instructions of that width from a 16 KB LDS image, for comparison
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x)                                                            \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

constexpr int ITER = 512;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename V>
__device__ __forceinline__ float hsum(V v);
template <>
__device__ __forceinline__ float hsum<float>(float v) { return v; }
template <>
__device__ __forceinline__ float hsum<f2>(f2 v) { return v.x + v.y; }
template <>
__device__ __forceinline__ float hsum<f4>(f4 v) { return v.x + v.y + v.z + v.w; }

template <typename V, int PATTERN>
__global__ __launch_bounds__(256) void k_global(const char *__restrict__ tab, float *out, unsigned mask) {
    const int lane = threadIdx.x & 63;
    constexpr unsigned W = sizeof(V);
    unsigned off;
    if (PATTERN == 0) off = lane * W;
    else if (PATTERN == 1) off = (lane & 15) * W + (lane >> 4) * 2176u;
    else off = lane * W + 4u;
    off += (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096u;
    float acc = 0.0f;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned o = (off + (unsigned)(i * 8 + j) * 16384u) & mask;
            asm volatile("" : "+v"(o));
            acc += hsum(*reinterpret_cast<const V *>(tab + o));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename V>
__global__ __launch_bounds__(256) void k_lds(const char *__restrict__ tab, float *out, unsigned mask) {
    __shared__ __attribute__((aligned(16))) char img[16384];
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float *>(img)[i] = reinterpret_cast<const float *>(tab)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    constexpr unsigned W = sizeof(V);
    unsigned off = lane * W + (threadIdx.x >> 6) * 1024u;
    float acc = 0.0f;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned o = (off + (unsigned)(i * 8 + j) * 2048u) & (16383u & ~(W - 1));
            asm volatile("" : "+v"(o));
            acc += hsum(*reinterpret_cast<const V *>(img + o));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;
    const size_t tab_bytes = 2u << 20;
    char *tab;
    float *out;
    CHECK(hipMalloc(&tab, tab_bytes + 65536));
    CHECK(hipMemset(tab, 0, tab_bytes + 65536));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Row { const char *name; void (*k)(const char *, float *, unsigned); int width; };
    const unsigned mask = (unsigned)(tab_bytes - 1) & ~15u;
    Row rows[] = {
        {"global dword   row", k_global<float, 0>, 4},   {"global dwordx2 row", k_global<f2, 0>, 8},
        {"global dwordx4 row", k_global<f4, 0>, 16},     {"global dword   tile16", k_global<float, 1>, 4},
        {"global dwordx2 tile16", k_global<f2, 1>, 8},   {"global dwordx4 tile16", k_global<f4, 1>, 16},
        {"global dword   shift1", k_global<float, 2>, 4}, {"global dwordx2 shift1", k_global<f2, 2>, 8},
        {"global dwordx4 shift1", k_global<f4, 2>, 16},  {"lds    b32", k_lds<float>, 4},
        {"lds    b64", k_lds<f2>, 8},                    {"lds    b128", k_lds<f4>, 16},
    };
    for (auto &r : rows) {
        hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, tab, out, mask);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, tab, out, mask);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_cu = 32.0 * ITER * 8 * 3;   // wave-instructions per CU (8 blocks x 4 waves)
        const double ns = ms * 1e6 / per_cu;
        printf("%-24s %8.3f ms  %6.2f ns per wave-instruction per CU = %5.1f cycles at 2.4 GHz, %6.1f B/clk/CU\n", r.name, ms, ns,
               ns * 2.4, 64.0 * r.width / (ns * 2.4));
    }
    return 0;
}
