// Dispatcher of the split-bf16 interval LSTM (kernel: lstm_split_kernel.h; instantiations: lstm_split_d*.hip).
#include "lstm_split_kernel.h"

namespace sagnn {

bool lstm_split_supported(int d) { return d == 32 || d == 64; }

int lstm_fwd_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W,
                   const float* b, float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out,
                   float* c_out, const float* h_init, int64_t ld_hi, const float* c_init, float* c_final,
                   hipStream_t s) {
  const bool save = gates_out != nullptr;
  // the tile's rows are addressed with 32-bit byte offsets from a per-tile base
  if (ld_h >= (1 << 22) || (int64_t)t * d >= (1 << 18))
    return fail(SAGNN_ERR_ARG, "split LSTM: output row stride must stay below 2^22 floats and t*d below 2^18");
  if (ld_h < (int64_t)t * d) return fail(SAGNN_ERR_ARG, "split LSTM: ld_h = %lld < t*d", (long long)ld_h);
  if (save && drop) return fail(SAGNN_ERR_ARG, "split LSTM: the training forward takes no dropout mask");
  if (d == 64) return save ? lstm_split_d64_save(SAGNN_LSTM_SPLIT_PASS) : lstm_split_d64(SAGNN_LSTM_SPLIT_PASS);
  if (d == 32) return save ? lstm_split_d32_save(SAGNN_LSTM_SPLIT_PASS) : lstm_split_d32(SAGNN_LSTM_SPLIT_PASS);
  return fail(SAGNN_ERR_DIM, "split LSTM supports d = 32 or 64, got %d", d);
}

}  // namespace sagnn
