// Split-f16 interval LSTM, d = 64, inference (with and without an output-dropout mask).
#include "lstm_f16_kernel.h"

namespace sagnn {
int lstm_f16_d64(SAGNN_LSTM_F16_ARGS) {
  if (drop) return launch_lstm_f16<64, false, true>(SAGNN_LSTM_F16_PASS);
  return launch_lstm_f16<64, false, false>(SAGNN_LSTM_F16_PASS);
}
}  // namespace sagnn
