// Split-bf16 interval LSTM, d = 64, inference (with and without an output-dropout mask).
#include "lstm_split_kernel.h"

namespace sagnn {
int lstm_split_d64(SAGNN_LSTM_SPLIT_ARGS) {
  if (drop) return launch_lstm_split<64, false, true>(SAGNN_LSTM_SPLIT_PASS);
  return launch_lstm_split<64, false, false>(SAGNN_LSTM_SPLIT_PASS);
}
}  // namespace sagnn
