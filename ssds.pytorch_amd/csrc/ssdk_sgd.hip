// ssdk_sgd.hip -- the optimizer update of the training step: SGD with momentum, weight decay and Nesterov, every parameter
// tensor of the model in a handful of launches, with the NaN/Inf skip decided on the DEVICE.
//
// Reference: optimizer.step() of the reference's loop (pipeline_anchor_apex.py:128-130) on torch.optim.SGD built by
// core/optimizer.py:73-134 (momentum 0.9, weight decay 1e-4), skipped when the loss is not finite (:110-111, 126-127: a
// host-side `continue` after .item() reads).  Rounds 4-5 used torch's fused multi-tensor SGD with its `found_inf` hook.
//
//     g   = grad + weight_decay * p
//     buf = momentum * buf + g                  (buffers are created as zeros: the first step then is buf = g, torch's rule)
//     p  -= lr * (nesterov ? g + momentum * buf : buf)
//
// A launch carries up to 40 tensors as kernel arguments (pointers + element counts + their first 4096-element block); a block
// finds its tensor by a linear walk over <= 40 prefix sums held in SGPRs, and updates 4096 consecutive elements with 16-byte
// accesses (p and buf read + written, grad read: 20 bytes per element -- an HBM stream).  `lr` may live on the device (a float
// the caller updates in place: a captured hipGraph keeps a LIVE learning rate), `found_inf` != 0 makes every block return
// before it touches anything.
#include "ssdk_common.h"

namespace ssdk {

constexpr int kSgdTensors = 40;
constexpr unsigned kSgdChunk = 4096;  // elements per block

struct SgdArgs {
  float* p[kSgdTensors];
  const float* g[kSgdTensors];
  float* m[kSgdTensors];
  unsigned n[kSgdTensors];
  unsigned start[kSgdTensors + 1];  // first block of tensor i
  int count;
  const float* lr_dev;
  float lr, momentum, weight_decay;
  int nesterov;
  const float* found_inf;
};

typedef float sgd_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sgd_momentum_kernel(const SgdArgs a) {
  if (a.found_inf && *a.found_inf != 0.f) return;  // the collective skip flag (pipeline_anchor_ddp.train_step)
  int t = 0;
  for (int i = 1; i < a.count; ++i)
    if (blockIdx.x >= a.start[i]) t = i;
  const unsigned base = (blockIdx.x - a.start[t]) * kSgdChunk, n = a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  const float lr = a.lr_dev ? *a.lr_dev : a.lr, mom = a.momentum, wd = a.weight_decay;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) == 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned i0 = base + ((unsigned)r * 256u + threadIdx.x) * 4u;
    if (i0 >= n) continue;
    if (vec && i0 + 3u < n) {
      sgd_f4 pv = *reinterpret_cast<const sgd_f4*>(p + i0);
      const sgd_f4 gv = *reinterpret_cast<const sgd_f4*>(g + i0);
      sgd_f4 mv = m ? *reinterpret_cast<const sgd_f4*>(m + i0) : sgd_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = gv[e] + wd * pv[e];
        float d = gg;
        if (m) {
          mv[e] = mom * mv[e] + gg;
          d = a.nesterov ? gg + mom * mv[e] : mv[e];
        }
        pv[e] = pv[e] - lr * d;
      }
      *reinterpret_cast<sgd_f4*>(p + i0) = pv;
      if (m) *reinterpret_cast<sgd_f4*>(m + i0) = mv;
    } else {
      for (unsigned i = i0; i < n && i < i0 + 4u; ++i) {
        const float gg = g[i] + wd * p[i];
        float d = gg;
        if (m) {
          const float mv = mom * m[i] + gg;
          m[i] = mv;
          d = a.nesterov ? gg + mom * mv : mv;
        }
        p[i] = p[i] - lr * d;
      }
    }
  }
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_sgd_step(int n, void* const* params, const void* const* grads, void* const* momentum_bufs, const int64_t* numel,
                             const float* lr_dev, float lr, float momentum, float weight_decay, int nesterov, const float* found_inf,
                             void* stream) {
  if (n < 0 || (n > 0 && (!params || !grads || !numel)) || (momentum != 0.f && n > 0 && !momentum_bufs)) {
    set_error("ssdk_sgd_step: bad argument");
    return SSDK_E_BADARG;
  }
  int i = 0;
  while (i < n) {
    SgdArgs a;
    a.count = 0;
    a.lr_dev = lr_dev;
    a.lr = lr;
    a.momentum = momentum;
    a.weight_decay = weight_decay;
    a.nesterov = nesterov;
    a.found_inf = found_inf;
    unsigned blocks = 0;
    for (; i < n && a.count < kSgdTensors; ++i) {
      if (numel[i] <= 0) continue;
      if (!params[i] || !grads[i] || (momentum != 0.f && !momentum_bufs[i]) || numel[i] >= ((int64_t)1 << 32)) {
        set_error("ssdk_sgd_step: tensor %d: null pointer or too large", i);
        return SSDK_E_BADARG;
      }
      a.p[a.count] = (float*)params[i];
      a.g[a.count] = (const float*)grads[i];
      a.m[a.count] = momentum != 0.f ? (float*)momentum_bufs[i] : nullptr;
      a.n[a.count] = (unsigned)numel[i];
      a.start[a.count] = blocks;
      blocks += ((unsigned)numel[i] + kSgdChunk - 1) / kSgdChunk;
      ++a.count;
    }
    if (a.count == 0) break;
    a.start[a.count] = blocks;
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    int rc = check_launch("sgd_momentum_kernel");
    if (rc) return rc;
  }
  return SSDK_OK;
}
