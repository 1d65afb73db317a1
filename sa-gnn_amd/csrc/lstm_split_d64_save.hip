// Split-bf16 interval LSTM, d = 64, training forward (stores gate activations and cell states).
#include "lstm_split_kernel.h"

namespace sagnn {
int lstm_split_d64_save(SAGNN_LSTM_SPLIT_ARGS) { return launch_lstm_split<64, true, false>(SAGNN_LSTM_SPLIT_PASS); }
}  // namespace sagnn
