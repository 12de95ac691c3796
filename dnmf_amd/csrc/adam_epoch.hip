// Adam on beta for a whole epoch of mini-batches in two launches.
//
// The reference hands ONE leaf tensor beta (10,3,T) to the caller's torch.optim.Adam (demo.py:42) and calls
// optimizer.step() once per mini-batch (Demix/dNMF.py:186-191).  The gradient of a mini-batch is non-zero
// only in the columns of its own frames, but every step still moves every column that has moment history
// ("coasting").  Column t therefore sees, in an epoch of n steps in which its mini-batch is number j:
//   j zero-gradient steps, one step with its gradient (evaluated at the coasted beta), n-1-j zero-gradient steps.
// Columns are independent, so the epoch is: phase 0 (coast j steps) -> K2 for all frames -> phase 1 (the
// gradient step and the remaining coasting).  Arithmetic follows torch's Adam (non-amsgrad, no weight
// decay): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*exp_avg_sq + (1-b2) g^2; bias corrections 1 - b^step in double,
// applied as fp32 scalars; denom = sqrt(v)/sqrt(bc2) + eps; p += (-lr/bc1) * (m/denom).
#include "common.hpp"

namespace dnmf {

// The step with a gradient, literally torch's: step_size = -(lr / (1 - b1^step)) and bc2_sqrt = sqrt(1 - b2^step) are
// evaluated in double and rounded to fp32 as torch does.
__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, float step_size, float bc2_sqrt,
                                         float b2f, float omb1, float omb2, float epsf) {
    m = m + omb1 * (g - m);                    // lerp_(grad, 1 - beta1)
    v = b2f * v + omb2 * (g * g);              // mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + epsf;
    p = p + step_size * (m / denom);           // addcdiv_(exp_avg, denom, value=-step_size)
}

constexpr int ADAM_CHUNK = 2048;  // steps whose scalars are tabulated in LDS at a time

// One thread per (coefficient e, frame t); with `order` the threads of a wave take frames whose mini-batches are
// neighbours in the epoch, so their step windows coincide.
//   phase 0: the j_t zero-gradient ("coasting") steps step0+1 .. step0+j_t before the frame's own mini-batch;
//   phase 1: step step0+j_t+1 with grad, then coasting up to step0+nsteps (frame_step[t] < 0: the frame is in no
//            mini-batch of this epoch and coasts through all nsteps here).
// The step with a gradient follows torch's fp32 arithmetic literally.  A run of coasting steps from state (p, m, v)
// has the closed form m_i = m b1^i, v_i = v b2^i, p += sum_i step_size_i m_i / (sqrt(v_i) / bc2_sqrt_i + eps): the
// terms are independent, so they are evaluated without the serial dependence of the step-by-step form (powers as
// running products in double, hardware sqrt / reciprocal, the sum in double) and the loop ends as soon as m_i
// underflows to zero (b1 = 0.9: after ~900 steps), which keeps the cost of an epoch independent of how many
// mini-batches the whole (sharded) video has.  Against torch's step-by-step fp32 evaluation the difference is below
// 1e-6 of the displacement -- less than the rounding torch itself accumulates by adding thousands of ~1e-5 increments
// to an fp32 parameter.
__global__ __launch_bounds__(256) void adam_epoch_kernel(float *__restrict__ beta, const float *__restrict__ grad,
                                                         float *__restrict__ m_, float *__restrict__ v_, int T,
                                                         long step0, const int *__restrict__ frame_step,
                                                         const int *__restrict__ order, int nsteps, double lr, double b1,
                                                         double b2, double eps, int phase) {
    __shared__ float s_step[ADAM_CHUNK], s_rbc2[ADAM_CHUNK], s_bc2[ADAM_CHUNK];
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = tid < 30L * T;
    const int e = live ? (int)(tid / T) : 0, rnk = live ? (int)(tid % T) : 0;
    const int t = order ? order[rnk] : rnk;
    const long i = (long)e * T + t;
    const int j = live ? frame_step[t] : 0;
    float p = 0.0f, m = 0.0f, v = 0.0f;
    if (live) p = beta[i], m = m_[i], v = v_[i];
    const float b2f = (float)b2, omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2), epsf = (float)eps;
    int first = 1, last = 0;  // 1-based step numbers inside the epoch
    if (live) {
        if (phase == 0)
            first = 1, last = j < 0 ? 0 : j;
        else
            first = j < 0 ? 1 : j + 1, last = nsteps;
    }
    const float gval = (live && phase == 1 && j >= 0) ? grad[i] : 0.0f;
    // coasting run in progress: state (m0, v0) at its start, running powers, the sum of its terms
    float m0 = m, v0 = v;
    double pb1 = 1.0, pb2 = 1.0, acc = 0.0;
    int run = 0;         // steps of the run so far
    bool spent = false;  // m has underflowed: the remaining terms of the run are exactly zero
    for (int c0 = 1; c0 <= nsteps; c0 += ADAM_CHUNK) {
        const int c1 = min(c0 + ADAM_CHUNK - 1, nsteps);
        const bool idle = first > last || c0 > last || spent;
        if (__syncthreads_and(idle)) break;  // nobody in the block has a term left in this or a later chunk
        for (int s = c0 + (int)threadIdx.x; s <= c1; s += blockDim.x) {
            const double st = (double)(step0 + s);
            const double bc2 = sqrt(1.0 - pow(b2, st));
            s_step[s - c0] = (float)(-(lr / (1.0 - pow(b1, st))));
            s_bc2[s - c0] = (float)bc2;
            s_rbc2[s - c0] = (float)(1.0 / bc2);
        }
        __syncthreads();
        int s = max(first, c0);
        const int s_end = min(last, c1);
        if (phase == 1 && j >= 0 && s == j + 1 && s <= s_end) {  // the step with the gradient opens the window
            adam_one(p, m, v, gval, s_step[s - c0], s_bc2[s - c0], b2f, omb1, omb2, epsf);
            m0 = m, v0 = v;
            ++s;
        }
        if (!spent) {
            for (; s <= s_end; ++s) {
                pb1 *= b1, pb2 *= b2;
                ++run;
                const float ms = m0 * (float)pb1;
                if (ms == 0.0f) {
                    spent = true;
                    break;
                }
                const float vs = v0 * (float)pb2;
                const float denom = fmaf(__builtin_amdgcn_sqrtf(vs), s_rbc2[s - c0], epsf);
                acc += (double)(s_step[s - c0] * (ms * __builtin_amdgcn_rcpf(denom)));
            }
        }
    }
    if (live && first <= last) {
        const int k = last - first + 1 - ((phase == 1 && j >= 0) ? 1 : 0);  // coasting steps of this phase
        if (k > 0) {
            p = (float)((double)p + acc);
            m = spent ? 0.0f : m0 * (float)pb1;
            v = v0 * (float)(run == k ? pb2 : pow(b2, (double)k));
        }
        beta[i] = p, m_[i] = m, v_[i] = v;
    }
}

}  // namespace dnmf

extern "C" int dnmf_adam_epoch(float *beta, const float *grad, float *exp_avg, float *exp_avg_sq, int T, long step0,
                               const int *frame_step, const int *order, int nsteps, double lr, double beta1,
                               double beta2, double eps, int phase, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(beta && exp_avg && exp_avg_sq && frame_step && (phase == 0 || grad), DNMF_E_NULL,
                 "dnmf_adam_epoch: NULL buffer");
    DNMF_REQUIRE(T > 0 && nsteps > 0 && step0 >= 0 && (phase == 0 || phase == 1), DNMF_E_SHAPE,
                 "dnmf_adam_epoch: T=%d nsteps=%d step0=%ld phase=%d", T, nsteps, step0, phase);
    const long n = 30L * T;
    hipLaunchKernelGGL(adam_epoch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, beta,
                       grad, exp_avg, exp_avg_sq, T, step0, frame_step, order, nsteps, lr, beta1, beta2, eps, phase);
    return check_launch("dnmf_adam_epoch");
}
