// Dynamic loss scaling for the fp16-operand mode of csrc/conv_np.hip, kept on the device so that the whole training step
// stays capturable in a hipGraph (no host read of the overflow flag).
//
// Reference behaviour: models/loss_collector.py:221-224 scales the loss with apex (`amp.scale_loss(loss, optimizer,
// loss_id)`), one scaler per loss (models/models.py:24-26 `num_losses=2`).  apex's rule: start at 2^16; after the backward
// pass un-scale the gradients; if any is inf / nan skip the optimiser step and halve the scale, otherwise step and, after
// `window` (2000) consecutive good steps, double it (cap 2^24).
//
// scaler = {scale, good_steps, found_inf, window, max_scale, min_scale} (fp32, device).  One step is
//   fsv_amp_check   : found_inf |= any(!finite(grad))             (after the gradient exchange, so every rank agrees)
//   fsv_amp_adam    : Adam on the flat buffers with grad * gscale / scale, a no-op (incl. the step counter) when found_inf
//   fsv_amp_update  : the scale rule above; clears found_inf
#include "fsv_common.h"

__global__ __launch_bounds__(256) void fsv_amp_check_kernel(const float* grad, long long n, float* scaler) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  bool bad = false;
  for (; i < n; i += stride) {
    const float g = grad[i];
    bad = bad || !(fabsf(g) <= 3.402823466e38f);        // false for inf and nan
  }
  if (bad) scaler[2] = 1.f;                              // benign race: every writer stores the same value
}

__global__ void fsv_amp_tick_kernel(float* state, const float* scaler, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && scaler[2] == 0.f) {
    float t = state[0] + 1.f;
    state[0] = t;
    state[1] = 1.f - powf(beta1, t);
    state[2] = 1.f - powf(beta2, t);
  }
}

// same update as fsv_adam_kernel (csrc/elementwise.hip; torch.optim.Adam as used at models/base_model.py:39-48)
__global__ __launch_bounds__(256) void fsv_amp_adam_kernel(float* param, const float* grad, float* m, float* v,
                                                           const float* state, const float* scaler, long long n,
                                                           float beta1, float beta2, float eps, float gscale) {
  if (scaler[2] != 0.f) return;
  const float bc1 = state[1], bc2 = state[2], lr = state[3];
  const float step_size = lr / bc1;
  const float rbc2 = 1.f / sqrtf(bc2);
  const float gs = gscale / scaler[0];
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n; i += stride) {
    float g = grad[i] * gs;
    float mi = beta1 * m[i] + (1.f - beta1) * g;
    float vi = beta2 * v[i] + (1.f - beta2) * g * g;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) * rbc2 + eps;
    param[i] = param[i] - step_size * (mi / denom);
  }
}

__global__ void fsv_amp_update_kernel(float* scaler) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float scale = scaler[0], good = scaler[1];
    if (scaler[2] != 0.f) {
      scale = fmaxf(scale * 0.5f, scaler[5]);
      good = 0.f;
    } else {
      good += 1.f;
      if (good >= scaler[3]) { scale = fminf(scale * 2.f, scaler[4]); good = 0.f; }
    }
    scaler[0] = scale; scaler[1] = good; scaler[2] = 0.f;
  }
}

static inline int fsv_amp_grid(long long n) {
  long long g = (n + 256 * 8 - 1) / (256 * 8);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

int fsv_amp_check(const float* grad, long long n, float* scaler, hipStream_t stream) {
  if (!grad || !scaler || n < 0) return FSV_ERR_BAD_ARG;
  if (n == 0) return FSV_OK;
  FSV_LAUNCH(fsv_amp_check_kernel, dim3(fsv_amp_grid(n)), dim3(256), stream, grad, n, scaler);
  return fsv_check_launch();
}

int fsv_amp_adam(float* param, const float* grad, float* m, float* v, float* state, float* scaler, long long n,
                 float beta1, float beta2, float eps, float gscale, hipStream_t stream) {
  if (!param || !grad || !m || !v || !state || !scaler || n < 0) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_amp_tick_kernel, dim3(1), dim3(64), stream, state, (const float*)scaler, beta1, beta2);
  FSV_LAUNCH(fsv_amp_adam_kernel, dim3(fsv_amp_grid(n / 4 + 1)), dim3(256), stream, param, grad, m, v, (const float*)state,
             (const float*)scaler, n, beta1, beta2, eps, gscale);
  return fsv_check_launch();
}

int fsv_amp_update(float* scaler, hipStream_t stream) {
  if (!scaler) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_amp_update_kernel, dim3(1), dim3(64), stream, scaler);
  return fsv_check_launch();
}

}  // extern "C"
