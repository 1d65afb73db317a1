// Split-bf16 interval LSTM, d = 32, training forward (stores gate activations and cell states).
#include "lstm_split_kernel.h"

namespace sagnn {
int lstm_split_d32_save(SAGNN_LSTM_SPLIT_ARGS) { return launch_lstm_split<32, true, false>(SAGNN_LSTM_SPLIT_PASS); }
}  // namespace sagnn
