// Issue-rate probe for the vector instructions the coordinate sequence of K2 / K3n is made of (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe.bin && tools/valu_probe.bin
// Every kernel runs ITER iterations of 8 independent chains of one instruction, 8 waves per SIMD on every CU; the
// figure printed is wave-instructions per nanosecond per SIMD relative to v_fma_f32 (1.00 = full rate).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));       \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int ITER = 4096;

#define PROBE(name, DECL, BODY, SINK)                                              \
    __global__ __launch_bounds__(256) void name(float *out, float s) {             \
        DECL;                                                                      \
        for (int i = 0; i < ITER; ++i) {                                           \
            BODY BODY BODY BODY BODY BODY BODY BODY                                \
        }                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = SINK;                                \
    }

// one asm statement = 8 independent instructions
#define F8(ins) asm volatile(ins " %0, %8, %9, %0\n" ins " %1, %8, %9, %1\n" ins " %2, %8, %9, %2\n" ins " %3, %8, %9, %3\n" \
                             ins " %4, %8, %9, %4\n" ins " %5, %8, %9, %5\n" ins " %6, %8, %9, %6\n" ins " %7, %8, %9, %7" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#define F8_2(ins) asm volatile(ins " %0, %8, %0\n" ins " %1, %8, %1\n" ins " %2, %8, %2\n" ins " %3, %8, %3\n" \
                               ins " %4, %8, %4\n" ins " %5, %8, %5\n" ins " %6, %8, %6\n" ins " %7, %8, %7" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
#define F8_1(ins) asm volatile(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" \
                               ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define F8_DPP(ins) asm volatile(ins " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" ins " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                                 ins " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n" ins " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                                 ins " %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n" ins " %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
                                 ins " %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n" ins " %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

#define DECLF float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, x = s, y = s * 0.5f
#define DECLI unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, x = (unsigned)s, y = x + 3
#define DECLP                                                                                                          \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                              \
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,      \
       a6 = a0 + 6.f, a7 = a0 + 7.f, x = {s, s}, y = {s * 0.5f, s}
#define SINKF (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define SINKI (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define SINKP (a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a7.y)

PROBE(k_fma, DECLF, F8("v_fma_f32"), SINKF)
PROBE(k_pk_fma, DECLP, F8("v_pk_fma_f32"), SINKP)
PROBE(k_pk_mul, DECLP, F8_2("v_pk_mul_f32"), SINKP)
PROBE(k_pk_add, DECLP, F8_2("v_pk_add_f32"), SINKP)
PROBE(k_mul, DECLF, F8_2("v_mul_f32"), SINKF)
PROBE(k_add, DECLF, F8_2("v_add_f32"), SINKF)
PROBE(k_mul_lo, DECLI, F8_2("v_mul_lo_u32"), SINKI)
PROBE(k_mad24, DECLI, F8("v_mad_u32_u24"), SINKI)
PROBE(k_mad_u64, DECLI, F8_2("v_mul_u32_u24"), SINKI)
PROBE(k_add_u32, DECLI, F8_2("v_add_u32"), SINKI)
PROBE(k_lshl_add, DECLI, F8("v_lshl_add_u32"), SINKI)
PROBE(k_floor, DECLF, F8_1("v_floor_f32"), SINKF)
PROBE(k_cvt_i, DECLF, F8_1("v_cvt_i32_f32"), SINKF)
PROBE(k_cvt_f, DECLF, F8_1("v_cvt_f32_i32"), SINKF)
PROBE(k_med3, DECLF, F8("v_med3_f32"), SINKF)
PROBE(k_med3i, DECLI, F8("v_med3_i32"), SINKI)
PROBE(k_min3i, DECLI, F8("v_min3_i32"), SINKI)
PROBE(k_rcp, DECLF, F8_1("v_rcp_f32"), SINKF)
PROBE(k_add_dpp, DECLF, F8_DPP("v_add_f32_dpp"), SINKF)
PROBE(k_mov, DECLF, F8_1("v_mov_b32"), SINKF)

// v_cmp + v_cndmask pair
__global__ __launch_bounds__(256) void k_cmp_cnd(float *out, float s) {
    DECLF;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %9, %0, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %9, %1, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %9, %2, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %9, %3, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %4\n v_cndmask_b32 %4, %9, %4, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %5\n v_cndmask_b32 %5, %9, %5, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %6\n v_cndmask_b32 %6, %9, %6, vcc\n"
                         "v_cmp_lt_f32 vcc, %8, %7\n v_cndmask_b32 %7, %9, %7, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x), "v"(y)
                         : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = SINKF;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    float *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Row { const char *name; void (*k)(float *, float); double per_iter; };
    std::vector<Row> rows = {
        {"v_fma_f32", k_fma, 64},       {"v_pk_fma_f32", k_pk_fma, 64}, {"v_pk_mul_f32", k_pk_mul, 64},
        {"v_pk_add_f32", k_pk_add, 64}, {"v_mul_f32", k_mul, 64},       {"v_add_f32", k_add, 64},
        {"v_mul_lo_u32", k_mul_lo, 64}, {"v_mad_u32_u24", k_mad24, 64}, {"v_mul_u32_u24", k_mad_u64, 64},
        {"v_add_u32", k_add_u32, 64},   {"v_lshl_add_u32", k_lshl_add, 64}, {"v_floor_f32", k_floor, 64},
        {"v_cvt_i32_f32", k_cvt_i, 64}, {"v_cvt_f32_i32", k_cvt_f, 64}, {"v_med3_f32", k_med3, 64},
        {"v_med3_i32", k_med3i, 64},    {"v_min3_i32", k_min3i, 64},    {"v_rcp_f32", k_rcp, 64},
        {"v_add_f32_dpp", k_add_dpp, 64}, {"v_mov_b32", k_mov, 64},     {"v_cmp+v_cndmask (pairs)", k_cmp_cnd, 32},
    };
    double base = 0;
    for (auto &r : rows) {
        hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);  // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per SIMD: 8 waves per SIMD x ITER x per_iter, 5 launches
        const double winst = 8.0 * ITER * r.per_iter * 5;
        const double rate = winst / (ms * 1e6);  // per ns per SIMD
        if (base == 0) base = rate;
        printf("%-26s %8.3f ms  %6.3f wave-instr/ns/SIMD  rel %5.2f  (cycles per wave-instr at 2.4 GHz: %.2f)\n", r.name, ms,
               rate, rate / base, 2.4 / rate);
    }
    return 0;
}
