// Dispatcher of the split-f16 interval LSTM (kernel: lstm_f16_kernel.h; instantiations: lstm_f16_d*.hip).
#include "lstm_f16_kernel.h"

namespace sagnn {

bool lstm_f16_supported(int d) { return d == 32 || d == 64; }

int lstm_fwd_f16(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W, const float* b,
                 float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out, float* c_out,
                 const float* h_init, int64_t ld_hi, const float* c_init, float* c_final, hipStream_t s) {
  const bool save = gates_out != nullptr;
  // the tile's rows are addressed with 32-bit byte offsets from a per-tile base
  if (ld_h >= (1 << 22) || (int64_t)t * d >= (1 << 18))
    return fail(SAGNN_ERR_ARG, "f16 LSTM: output row stride must stay below 2^22 floats and t*d below 2^18");
  if (ld_h < (int64_t)t * d) return fail(SAGNN_ERR_ARG, "f16 LSTM: ld_h = %lld < t*d", (long long)ld_h);
  if (save && drop) return fail(SAGNN_ERR_ARG, "f16 LSTM: the training forward takes no dropout mask");
  if (d == 64) return save ? lstm_f16_d64_save(SAGNN_LSTM_F16_PASS) : lstm_f16_d64(SAGNN_LSTM_F16_PASS);
  if (d == 32) return save ? lstm_f16_d32_save(SAGNN_LSTM_F16_PASS) : lstm_f16_d32(SAGNN_LSTM_F16_PASS);
  return fail(SAGNN_ERR_DIM, "f16 LSTM supports d = 32 or 64, got %d", d);
}

}  // namespace sagnn
