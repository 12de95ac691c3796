// Adam on beta for a whole epoch of mini-batches in two launches.
//
// The reference hands ONE leaf tensor beta (10,3,T) to the caller's torch.optim.Adam (demo.py:42) and calls
// optimizer.step() once per mini-batch (Demix/dNMF.py:186-191).  The gradient of a mini-batch is non-zero
// only in the columns of its own frames, but every step still moves every column that has moment history
// ("coasting").  Column t therefore sees, in an epoch of n steps in which its mini-batch is number j:
//   j zero-gradient steps, one step with its gradient (evaluated at the coasted beta), n-1-j zero-gradient steps.
// Columns are independent, so the epoch is: phase 0 (coast j steps) -> K2 for all frames -> phase 1 (the
// gradient step and the remaining coasting).  Arithmetic follows torch's Adam (non-amsgrad, no weight
// decay): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*exp_avg_sq + (1-b2) g^2; bias corrections in double,
// applied as fp32 scalars; denom = sqrt(v)/sqrt(bc2) + eps; p += (-lr/bc1) * (m/denom).
#include "common.hpp"

namespace dnmf {

__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, double pow1, double pow2, double lr,
                                         float b2f, float omb1, float omb2, float epsf) {
    m = m + omb1 * (g - m);                    // lerp_(grad, 1 - beta1)
    v = b2f * v + omb2 * (g * g);              // mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const double bc1 = 1.0 - pow1, bc2 = 1.0 - pow2;
    const float step_size = (float)(-(lr / bc1));
    const float bc2_sqrt = (float)sqrt(bc2);
    const float denom = sqrtf(v) / bc2_sqrt + epsf;
    p = p + step_size * (m / denom);           // addcdiv_(exp_avg, denom, value=-step_size)
}

// One thread per (coefficient e, frame t).  phase 0: steps step0+1 .. step0+j_t with zero gradient.
// phase 1: step step0+j_t+1 with grad, then zero-gradient steps up to step0+nsteps.  frame_step[t] < 0: the
// frame is in no mini-batch of this epoch -> it coasts through all nsteps (done in phase 1).
__global__ __launch_bounds__(256) void adam_epoch_kernel(float *__restrict__ beta, const float *__restrict__ grad,
                                                         float *__restrict__ m_, float *__restrict__ v_, int T,
                                                         long step0, const int *__restrict__ frame_step, int nsteps,
                                                         double lr, double b1, double b2, double eps, int phase) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 30L * T) return;
    const int t = (int)(i % T);
    const int j = frame_step[t];
    float p = beta[i], m = m_[i], v = v_[i];
    const float b2f = (float)b2, omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2), epsf = (float)eps;
    int first, last;  // 1-based step numbers inside the epoch
    if (phase == 0) {
        first = 1, last = j < 0 ? 0 : j;
    } else {
        first = j < 0 ? 1 : j + 1, last = nsteps;
    }
    if (first > last) return;
    double pow1 = pow(b1, (double)(step0 + first)), pow2 = pow(b2, (double)(step0 + first));
    for (int s = first; s <= last; ++s) {
        const float g = (phase == 1 && s == j + 1 && j >= 0) ? grad[i] : 0.0f;
        adam_one(p, m, v, g, pow1, pow2, lr, b2f, omb1, omb2, epsf);
        pow1 *= b1, pow2 *= b2;
    }
    beta[i] = p, m_[i] = m, v_[i] = v;
}

}  // namespace dnmf

extern "C" int dnmf_adam_epoch(float *beta, const float *grad, float *exp_avg, float *exp_avg_sq, int T, long step0,
                               const int *frame_step, int nsteps, double lr, double beta1, double beta2, double eps,
                               int phase, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(beta && exp_avg && exp_avg_sq && frame_step && (phase == 0 || grad), DNMF_E_NULL,
                 "dnmf_adam_epoch: NULL buffer");
    DNMF_REQUIRE(T > 0 && nsteps > 0 && step0 >= 0 && (phase == 0 || phase == 1), DNMF_E_SHAPE,
                 "dnmf_adam_epoch: T=%d nsteps=%d step0=%ld phase=%d", T, nsteps, step0, phase);
    const long n = 30L * T;
    hipLaunchKernelGGL(adam_epoch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, beta,
                       grad, exp_avg, exp_avg_sq, T, step0, frame_step, nsteps, lr, beta1, beta2, eps, phase);
    return check_launch("dnmf_adam_epoch");
}
