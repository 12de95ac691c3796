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
// applied as fp32 scalars; denom = sqrt(v)/sqrt(bc2) + eps; p += (-lr/bc1) * (m/denom) -- with the fused multiply-adds
// torch's GPU kernels contract these expressions into (tools/adam_probe.py compares one optimiser step of the
// installed torch with the candidate sequences on 10^6 random elements: this one matches on every element).
#include "common.hpp"

namespace dnmf {

constexpr int ADAM_LITERAL = 64;   // coasting runs up to this length are stepped one by one in torch's own arithmetic

// The step with a gradient, literally torch's: step_size = -(lr / (1 - b1^step)) and bc2_sqrt = sqrt(1 - b2^step) are
// evaluated in double and rounded to fp32 as torch does.
__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, float step_size, float bc2_sqrt,
                                         float b2f, float omb1, float omb2, float epsf) {
    m = fmaf(omb1, g - m, m);                  // lerp_(grad, 1 - beta1)
    v = fmaf(omb2, g * g, b2f * v);            // mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + epsf;
    p = fmaf(step_size, m / denom, p);         // addcdiv_(exp_avg, denom, value=-step_size)
}

// The step-dependent scalars of steps step0+1 .. step0+nsteps, the same for every column, as one float4 per step:
// x = -(lr / (1 - b1^step)), y = sqrt(1 - b2^step), z = 1 / y; double precision, rounded to fp32 once.
__global__ __launch_bounds__(256) void adam_table_kernel(float4 *__restrict__ tab, long step0, int nsteps, double lr,
                                                         double b1, double b2) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (s > nsteps) return;
    const double st = (double)(step0 + s);
    const double bc2 = sqrt(1.0 - pow(b2, st));
    tab[s - 1] = make_float4((float)(-(lr / (1.0 - pow(b1, st)))), (float)bc2, (float)(1.0 / bc2), 0.0f);
}

// One thread per (coefficient e, frame t); with `order` the threads of a wave take frames whose mini-batches are
// neighbours in the epoch, so their step windows coincide.
//   phase 0: the j_t zero-gradient ("coasting") steps step0+1 .. step0+j_t before the frame's own mini-batch;
//   phase 1: step step0+j_t+1 with grad, then coasting up to step0+nsteps (frame_step[t] < 0: the frame is in no
//            mini-batch of this epoch and coasts through all nsteps here).
// The step with a gradient follows torch's fp32 arithmetic literally, and so does a run of at most ADAM_LITERAL coasting
// steps (a zero gradient through the same expressions): an epoch of few mini-batches -- the reference's demo has 25 --
// is then the optimiser's own sequence bit for bit.  A longer run of coasting steps from state (p, m, v)
// has the closed form m_i = m b1^i, v_i = v b2^i, p += sum_i step_size_i m_i / (sqrt(v_i) / bc2_sqrt_i + eps): the
// terms are independent, so they are evaluated without the serial dependence of the step-by-step form (powers as
// running products in double, hardware sqrt / reciprocal, the sum in double) and the loop covers only the steps before
// m_i underflows to zero (b1 = 0.9: ~900 steps; the count is known up front, so the loop has no data-dependent exit
// and its table reads pipeline), which keeps the cost of an epoch independent of how many
// mini-batches the whole (sharded) video has.  Against torch's step-by-step fp32 evaluation the difference is below
// 1e-6 of the displacement -- less than the rounding torch itself accumulates by adding thousands of ~1e-5 increments
// to an fp32 parameter.
__global__ __launch_bounds__(256) void adam_epoch_kernel(float *__restrict__ beta, const float *__restrict__ grad,
                                                         float *__restrict__ m_, float *__restrict__ v_, int T,
                                                         const int *__restrict__ frame_step,
                                                         const int *__restrict__ order, int nsteps,
                                                         const float4 *__restrict__ tab, double b1, double b2,
                                                         double eps, int phase) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 30L * T) return;
    const int e = (int)(tid / T), rnk = (int)(tid % T);
    const int t = order ? order[rnk] : rnk;
    const long i = (long)e * T + t;
    const int j = frame_step[t];
    int first, last;  // 1-based step numbers inside the epoch
    if (phase == 0)
        first = 1, last = j < 0 ? 0 : j;
    else
        first = j < 0 ? 1 : j + 1, last = nsteps;
    if (first > last) return;
    float p = beta[i], m = m_[i], v = v_[i];
    const float b2f = (float)b2, omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2), epsf = (float)eps;
    int s = first;
    if (phase == 1 && j >= 0) {  // the step with the gradient opens the window
        const float4 q = tab[s - 1];
        adam_one(p, m, v, grad[i], q.x, q.y, b2f, omb1, omb2, epsf);
        ++s;
    }
    const int k = last - s + 1;  // coasting steps
    if (k > 0 && k <= ADAM_LITERAL) {
        for (int r = 0; r < k; ++r) {
            const float4 q = tab[s - 1 + r];
            adam_one(p, m, v, 0.0f, q.x, q.y, b2f, omb1, omb2, epsf);
        }
    } else if (k > 0) {
        // steps until m b1^i is below the smallest fp32 denormal (2^-149): every later term is exactly zero
        int live_steps = 0;
        if (m != 0.0f) {
            // (a denormal m may read as log2 = -inf: no live steps, which is what its product with b1 gives anyway)
            const float need = fmaxf((__log2f(fabsf(m)) + 150.0f) / -__log2f((float)b1), -1.0f);
            live_steps = (need >= (float)k || !(b1 < 1.0)) ? k : max(0, (int)need + 2);  // b1 >= 1: m never decays
            live_steps = min(live_steps, k);
        }
        double pb1 = 1.0, pb2 = 1.0, acc = 0.0;
#pragma unroll 4
        for (int r = 0; r < live_steps; ++r) {
            const float4 q = tab[s - 1 + r];
            pb1 *= b1, pb2 *= b2;
            const float ms = m * (float)pb1;
            const float vs = v * (float)pb2;
            const float denom = fmaf(__builtin_amdgcn_sqrtf(vs), q.z, epsf);
            acc += (double)(q.x * (ms * __builtin_amdgcn_rcpf(denom)));
        }
        p = (float)((double)p + acc);
        m = live_steps == k ? m * (float)pb1 : 0.0f;
        v = v * (float)(live_steps == k ? pb2 : pow(b2, (double)k));
    }
    beta[i] = p, m_[i] = m, v_[i] = v;
}

}  // namespace dnmf

extern "C" {

size_t dnmf_adam_epoch_workspace(int nsteps) { return nsteps > 0 ? (size_t)nsteps * 4 * sizeof(float) : 0; }

int dnmf_adam_epoch(float *beta, const float *grad, float *exp_avg, float *exp_avg_sq, int T, long step0,
                    const int *frame_step, const int *order, int nsteps, double lr, double beta1, double beta2,
                    double eps, int phase, void *workspace, size_t workspace_bytes, dnmf_stream_t stream) {
    using namespace dnmf;
    DNMF_REQUIRE(beta && exp_avg && exp_avg_sq && frame_step && workspace && (phase == 0 || grad), DNMF_E_NULL,
                 "dnmf_adam_epoch: NULL buffer");
    DNMF_REQUIRE(T > 0 && nsteps > 0 && step0 >= 0 && (phase == 0 || phase == 1), DNMF_E_SHAPE,
                 "dnmf_adam_epoch: T=%d nsteps=%d step0=%ld phase=%d", T, nsteps, step0, phase);
    DNMF_REQUIRE(workspace_bytes >= dnmf_adam_epoch_workspace(nsteps), DNMF_E_WORKSPACE,
                 "dnmf_adam_epoch: workspace %zu < %zu bytes", workspace_bytes, dnmf_adam_epoch_workspace(nsteps));
    DNMF_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, DNMF_E_SHAPE, "dnmf_adam_epoch: workspace must be 16-byte aligned");
    float4 *tab = static_cast<float4 *>(workspace);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_table_kernel, dim3((unsigned)((nsteps + 255) / 256)), dim3(256), 0, st, tab, step0, nsteps, lr,
                       beta1, beta2);
    const long n = 30L * T;
    hipLaunchKernelGGL(adam_epoch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, beta, grad, exp_avg,
                       exp_avg_sq, T, frame_step, order, nsteps, tab, beta1, beta2, eps, phase);
    return check_launch("dnmf_adam_epoch");
}

}  // extern "C"
